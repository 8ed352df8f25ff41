"""ctypes binding of libmlrt_hip.so (include/mlrt.h + include/mlrt_host.h) for tests and bench.py.

The names mirror the reference's interface for the path (src/worlds.rs `World`, src/camera.rs `Camera::new`,
src/raytrace.rs `RenderingParams`, `Renderer::new_with_rng(...).render()`).  All compute happens in the HIP
library; this module never falls back to a CPU implementation and raises if the library is missing.

Load with `importlib.import_module("mu-lambda-raytracer_amd")` (the directory name has hyphens).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLRT_LIB") or os.path.join(_HERE, "lib", "libmlrt_hip.so")  # MLRT_LIB: A/B builds
CLI_PATH = os.path.join(_HERE, "bin", "mlrt")
EARTHMAP_PPM = os.path.join(os.path.dirname(_HERE), "assets", "earthmap.ppm")

MLRT_OK, MLRT_ERR_INVALID, MLRT_ERR_NO_DEVICE, MLRT_ERR_HIP, MLRT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
PRECISION_FP32, PRECISION_FP64, PRECISION_FP64_STRICT = 0, 1, 2
GATHER_AUTO, GATHER_RCCL, GATHER_COPY = 0, 1, 2
BG_BLACK, BG_GRADIENT = 0, 1
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
PRIM_SPHERE, PRIM_BOX, PRIM_RECT = 0, 1, 2
OBJ_PRIM, OBJ_BVH, OBJ_MEDIUM = 0, 1, 2
BVH_EMPTY = -(2 ** 31)


class MlrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mlrt error %d: %s" % (code, message))
        self.code = code


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("index2", C.c_int32), ("_pad", C.c_int32), ("color", C.c_double * 3),
                ("scale", C.c_double)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class Perlin(C.Structure):
    _fields_ = [("ranvec", C.POINTER(C.c_double)), ("perm", C.POINTER(C.c_int32))]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class Box(C.Structure):
    _fields_ = [("min", C.c_double * 3), ("max", C.c_double * 3), ("material", C.c_int32), ("_pad", C.c_int32)]


class Rect(C.Structure):
    _fields_ = [("a0", C.c_int32), ("a1", C.c_int32), ("a0_v0", C.c_double), ("a0_v1", C.c_double), ("a1_v0", C.c_double),
                ("a1_v1", C.c_double), ("plane", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class PrimRef(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class BvhNode(C.Structure):
    _fields_ = [("bounds_min", C.c_double * 3), ("bounds_max", C.c_double * 3), ("left", C.c_int32), ("right", C.c_int32)]


class Bvh(C.Structure):
    _fields_ = [("root", C.c_int32), ("_pad", C.c_int32)]


class Transform(C.Structure):
    _fields_ = [("has_rotate", C.c_int32), ("axis", C.c_int32), ("sin_theta", C.c_double), ("cos_theta", C.c_double),
                ("has_translate", C.c_int32), ("_pad", C.c_int32), ("offset", C.c_double * 3)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("bvh", C.c_int32), ("prim", PrimRef), ("transform", C.c_int32), ("phase_material", C.c_int32),
                ("neg_inv_density", C.c_double)]


class Scene(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_spheres", "n_boxes", "n_rects", "n_prim_refs", "n_bvh_nodes", "n_bvhs", "n_transforms", "n_objects",
                                         "n_materials", "n_textures", "n_images", "n_perlins")] + [
        ("spheres", C.POINTER(Sphere)), ("boxes", C.POINTER(Box)), ("rects", C.POINTER(Rect)), ("prim_refs", C.POINTER(PrimRef)),
        ("bvh_nodes", C.POINTER(BvhNode)), ("bvhs", C.POINTER(Bvh)), ("transforms", C.POINTER(Transform)), ("objects", C.POINTER(Object)),
        ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)), ("images", C.POINTER(Image)), ("perlins", C.POINTER(Perlin))]


class CameraStruct(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("lens_radius", C.c_double)]


class Params(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("background_kind", C.c_int32), ("tracer", C.c_int32), ("background_top", C.c_double * 3),
                ("background_bottom", C.c_double * 3), ("light_source", C.c_double * 3), ("light_intensity", C.c_double)]


TRACER_RECURSIVE, TRACER_SINGLE_LIGHT = 0, 1


class Options(C.Structure):
    """mlrt_options (include/mlrt.h): tuning / diagnostic switches, none of which may change an f64 image."""
    _fields_ = [("sah", C.c_int32), ("leaf_wrap", C.c_int32), ("primary_lists", C.c_int32), ("lds_scene", C.c_int32), ("force_full", C.c_int32), ("cull_f64", C.c_int32),
                ("tile_samples", C.c_int32), ("tile_queues", C.c_int32), ("n_tail", C.c_int32), ("tail", C.c_int32 * 8), ("stats", C.c_int32),
                ("debug_pixel", C.c_int32), ("debug_sample", C.c_int32), ("stall_cap", C.c_int32), ("pair_walk", C.c_int32), ("split_roles", C.c_int32), ("single_light_tracer", C.c_int32)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int32, C.c_int32, C.c_void_p)

# every symbol declared in include/mlrt.h and include/mlrt_host.h
EXPORTS_MLRT_H = ["mlrt_version", "mlrt_last_error", "mlrt_device_count", "mlrt_device_warm_up", "mlrt_options_default", "mlrt_options_from_env", "mlrt_context_create",
                  "mlrt_context_create_opts", "mlrt_context_destroy", "mlrt_context_render",
                  "mlrt_context_render_device", "mlrt_context_check", "mlrt_context_last_kernel_ms", "mlrt_context_info", "mlrt_context_kernel_variant", "mlrt_render", "mlrt_rows_for_device", "mlrt_multi_create",
                  "mlrt_multi_destroy", "mlrt_multi_render", "mlrt_multi_info", "mlrt_render_multi"]
EXPORTS_MLRT_HOST_H = ["mlrt_world_count", "mlrt_world_name", "mlrt_world_build", "mlrt_world_free", "mlrt_world_scene", "mlrt_world_camera",
                       "mlrt_world_background", "mlrt_world_stream_probes", "mlrt_camera_new", "mlrt_image_height", "mlrt_write_p3",
                       "mlrt_write_p3_path", "mlrt_pcg64_stream", "mlrt_decode_jpeg", "mlrt_decode_jpeg_file", "mlrt_free"]

_lib = None


def lib():
    """The loaded HIP library.  Fails loudly when it has not been built: there is no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MlrtError(MLRT_ERR_UNSUPPORTED, "libmlrt_hip.so is missing at %s - run `python mu-lambda-raytracer_amd/build.py`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.mlrt_last_error.restype = C.c_char_p
        L.mlrt_world_name.restype = C.c_char_p
        L.mlrt_world_build.restype = C.c_void_p
        L.mlrt_world_build.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32]
        L.mlrt_world_free.argtypes = [C.c_void_p]
        L.mlrt_world_scene.restype = C.POINTER(Scene)
        L.mlrt_world_scene.argtypes = [C.c_void_p]
        L.mlrt_world_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mlrt_world_background.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.mlrt_world_stream_probes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.mlrt_camera_new.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + [C.POINTER(CameraStruct)]
        L.mlrt_image_height.argtypes = [C.c_int32, C.c_double]
        L.mlrt_write_p3_path.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.mlrt_pcg64_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p]
        L.mlrt_decode_jpeg.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mlrt_decode_jpeg_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mlrt_free.argtypes = [C.c_void_p]
        L.mlrt_context_create.argtypes = [C.POINTER(Scene), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.mlrt_context_create_opts.argtypes = [C.POINTER(Scene), C.c_int32, C.c_int32, C.POINTER(Options), C.POINTER(C.c_void_p)]
        L.mlrt_options_default.argtypes = [C.POINTER(Options)]
        L.mlrt_options_from_env.argtypes = [C.POINTER(Options)]
        L.mlrt_context_destroy.argtypes = [C.c_void_p]
        L.mlrt_context_render.argtypes = [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(Params), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mlrt_context_render_device.argtypes = [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(Params), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mlrt_context_check.argtypes = [C.c_void_p]
        L.mlrt_context_last_kernel_ms.restype = C.c_double
        L.mlrt_context_last_kernel_ms.argtypes = [C.c_void_p]
        L.mlrt_context_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.mlrt_context_kernel_variant.argtypes = [C.c_void_p]
        L.mlrt_render.argtypes = [C.POINTER(Scene), C.POINTER(CameraStruct), C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mlrt_rows_for_device.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.mlrt_multi_create.argtypes = [C.POINTER(Scene), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(Options), C.c_int32, C.POINTER(C.c_void_p)]
        L.mlrt_multi_destroy.argtypes = [C.c_void_p]
        L.mlrt_multi_render.argtypes = [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]
        L.mlrt_multi_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.mlrt_render_multi.argtypes = [C.POINTER(Scene), C.POINTER(CameraStruct), C.POINTER(Params), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != MLRT_OK:
        raise MlrtError(rc, lib().mlrt_last_error().decode())


def device_count():
    return int(lib().mlrt_device_count())


def world_names():
    L = lib()
    return [L.mlrt_world_name(i).decode() for i in range(L.mlrt_world_count())]


def load_earthmap(path=EARTHMAP_PPM):
    """The decoded earth texture (assets/earthmap.ppm, binary P6) as a uint8 array [H][W][3]."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = [int(x) for x in f.readline().split()]
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(w * h * 3), dtype=np.uint8).reshape(h, w, 3).copy()


def decode_jpeg(data):
    """Baseline JPEG bytes -> uint8 [H][W][3] through the library's own reader (mlrt_decode_jpeg)."""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    out, w, h = C.POINTER(C.c_uint8)(), C.c_int32(), C.c_int32()
    _check(lib().mlrt_decode_jpeg(buf, len(data), C.byref(out), C.byref(w), C.byref(h)))
    img = np.ctypeslib.as_array(out, shape=(h.value, w.value, 3)).copy()
    lib().mlrt_free(out)
    return img


def options(**kw):
    """mlrt_options_from_env() with fields overridden by keyword (tail=[...] sets n_tail too)."""
    o = Options()
    lib().mlrt_options_from_env(C.byref(o))
    for k, v in kw.items():
        if k == "tail":
            o.n_tail = len(v)
            for i, x in enumerate(v):
                o.tail[i] = int(x)
        else:
            if not hasattr(o, k):
                raise AttributeError("mlrt_options has no field %r" % k)
            setattr(o, k, int(v))
    return o


def image_height(image_width, aspect_ratio):
    return int(lib().mlrt_image_height(int(image_width), float(aspect_ratio)))


class World:
    """trait World (worlds.rs:14-19) + World::build(rng(0))."""

    def __init__(self, name, seed, earth=None):
        L = lib()
        if earth is None and name in ("earth", "final_scene"):
            earth = load_earthmap()
        self._earth = None if earth is None else np.ascontiguousarray(earth, dtype=np.uint8)
        ptr = None if self._earth is None else self._earth.ctypes.data_as(C.c_void_p)
        eh, ew = (0, 0) if self._earth is None else self._earth.shape[:2]
        self.handle = L.mlrt_world_build(name.encode(), int(seed), ptr, ew, eh)
        if not self.handle:
            raise MlrtError(MLRT_ERR_INVALID, L.mlrt_last_error().decode())
        self.name, self.seed = name, int(seed)

    @property
    def scene(self):
        return lib().mlrt_world_scene(self.handle)

    def camera(self):
        lf, la, fov = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        lib().mlrt_world_camera(self.handle, lf, la, C.byref(fov))
        return list(lf), list(la), fov.value

    def background(self, params):
        lib().mlrt_world_background(self.handle, C.byref(params))

    def stream_probes(self):
        out = (C.c_uint64 * 2)()
        lib().mlrt_world_stream_probes(self.handle, out)
        return int(out[0]), int(out[1])

    def close(self):
        if self.handle:
            lib().mlrt_world_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist):
    """Camera::new (camera.rs:15-38)."""
    out = CameraStruct()
    lib().mlrt_camera_new((C.c_double * 3)(*lookfrom), (C.c_double * 3)(*lookat), (C.c_double * 3)(*vup), float(vfov), float(aspect_ratio),
                          float(aperture), float(focus_dist), C.byref(out))
    return out


def make_params(world, image_width, image_height, samples_per_pixel, max_depth=50, seed=None, single_light=None):
    """single_light = (light_source xyz, intensity): SingleLightSourceRayTracer instead of RecursiveRayTracer{max_depth} (raytrace.rs:110-135)."""
    p = Params()
    p.image_width, p.image_height, p.samples_per_pixel, p.max_depth = int(image_width), int(image_height), int(samples_per_pixel), int(max_depth)
    p.seed = world.seed if seed is None else int(seed)
    world.background(p)
    if single_light is not None:
        p.tracer = TRACER_SINGLE_LIGHT
        for k in range(3):
            p.light_source[k] = float(single_light[0][k])
        p.light_intensity = float(single_light[1])
    return p


def default_camera(world, aspect_ratio, aperture=0.0, focus_dist=None, vup=(0.0, 1.0, 0.0)):
    """The camera main.rs builds from the world defaults (main.rs:104-112, 192-200)."""
    lookfrom, lookat, fov = world.camera()
    if focus_dist is None:
        focus_dist = float(np.sqrt(sum((a - b) * (a - b) for a, b in zip(lookat, lookfrom))))
    return camera_new(lookfrom, lookat, vup, fov, aspect_ratio, aperture, focus_dist)


class Renderer:
    """Renderer::new_with_rng(camera, world, background, params, RecursiveRayTracer{max_depth}, SeedableRngator(seed)) + render()."""

    def __init__(self, world, camera, params, precision=PRECISION_FP32, device=0, opts=None):
        self.world, self.camera, self.params, self.precision, self.device = world, camera, params, precision, device
        ctx = C.c_void_p()
        if opts is None and params.tracer == TRACER_SINGLE_LIGHT:
            opts = options(single_light_tracer=1)  # (the context must carry that tracer's kernel variant, include/mlrt.h)
        if opts is None:
            _check(lib().mlrt_context_create(world.scene, int(device), int(precision), C.byref(ctx)))
        else:
            _check(lib().mlrt_context_create_opts(world.scene, int(device), int(precision), C.byref(opts), C.byref(ctx)))
        self.ctx = ctx

    def render(self, rows=None, progress=None):
        """Returns int32 [n_rows][W][3]; row j = 0 is the bottom row (as Vec<Vec<RGB>>)."""
        p = self.params
        if rows is None:
            n, rows_ptr = p.image_height, None
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            n, rows_ptr = int(rows.size), rows.ctypes.data_as(C.c_void_p)
        out = np.zeros((n, p.image_width, 3), dtype=np.int32)
        cb = PROGRESS_FN(progress) if progress else None
        _check(lib().mlrt_context_render(self.ctx, C.byref(self.camera), C.byref(p), rows_ptr, n, out.ctypes.data_as(C.c_void_p),
                                         C.cast(cb, C.c_void_p) if cb else None, None))
        return out

    def render_device(self, d_rows_ptr, n_rows, d_rgb_ptr, d_accum_ptr=None, stream=None):
        """Asynchronous launch on device buffers (raw pointers, e.g. torch tensor.data_ptr())."""
        _check(lib().mlrt_context_render_device(self.ctx, C.byref(self.camera), C.byref(self.params), d_rows_ptr, int(n_rows), d_rgb_ptr, d_accum_ptr,
                                                stream))

    def check(self):
        """Waits for the last launch; raises if the render kernel's watchdog tripped."""
        _check(lib().mlrt_context_check(self.ctx))

    def last_kernel_ms(self):
        return float(lib().mlrt_context_last_kernel_ms(self.ctx))

    def info(self):
        out = (C.c_int64 * 4)()
        _check(lib().mlrt_context_info(self.ctx, out))
        return {"blob_bytes": int(out[0]), "lds_scene": bool(out[1]), "block_threads": int(out[2]), "grid": int(out[3]),
                "kernel_variant": (lambda v: {0: "full", 1: "lean-threaded", 2: "lean-pair"}.get(v, "full-subset-%d" % (v - 3)))(int(lib().mlrt_context_kernel_variant(self.ctx)))}

    def close(self):
        if getattr(self, "ctx", None):
            lib().mlrt_context_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rows_for_device(image_height, k, n_devices):
    """mlrt_rows_for_device: the rows device k of n renders (round-robin)."""
    out = (C.c_int32 * max(1, int(image_height)))()
    n = lib().mlrt_rows_for_device(int(image_height), int(k), int(n_devices), out)
    return [int(out[i]) for i in range(n)]


class MultiRenderer:
    """mlrt_multi_*: one frame over several GPUs from one process (one context and stream per device, one gather)."""

    def __init__(self, world, camera, params, devices, precision=PRECISION_FP64, gather=GATHER_AUTO, opts=None):
        self.world, self.camera, self.params, self.devices = world, camera, params, list(devices)
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        handle = C.c_void_p()
        _check(lib().mlrt_multi_create(world.scene, dev, len(self.devices), int(precision), C.byref(opts) if opts is not None else None, int(gather), C.byref(handle)))
        self.handle = handle

    def render(self, progress=None):
        p = self.params
        out = np.zeros((p.image_height, p.image_width, 3), dtype=np.int32)
        cb = PROGRESS_FN(progress) if progress else None
        _check(lib().mlrt_multi_render(self.handle, C.byref(self.camera), C.byref(p), out.ctypes.data_as(C.c_void_p), C.cast(cb, C.c_void_p) if cb else None, None))
        return out

    def info(self):
        frame_ms, gather = C.c_double(), C.c_int32()
        kernel_ms = (C.c_double * len(self.devices))()
        _check(lib().mlrt_multi_info(self.handle, C.byref(frame_ms), kernel_ms, C.byref(gather)))
        return {"frame_ms": frame_ms.value, "kernel_ms": list(kernel_ms), "gather": int(gather.value)}

    def close(self):
        if getattr(self, "handle", None):
            lib().mlrt_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
