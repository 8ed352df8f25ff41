"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
RNG_PCG_ROW, RNG_PHILOX_SAMPLE = 0, 1
STAT_NAMES = ["samples", "rays", "aabb_tests", "sphere_tests", "rect_tests", "scatters", "draws", "noise_evals", "texel_fetches", "medium_tests"]


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "oracle_scene.cpp", "oracle_core.hpp", "oracle_main.cpp", "Makefile")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B" if force else "-s", "all"])
    return LIB_PATH


class OracleWorld:
    def __init__(self, oracle, handle, name, seed):
        self.oracle, self.handle, self.name, self.seed = oracle, handle, name, seed

    def prims(self):
        L = self.oracle.L
        n = L.orc_world_prim_count(self.handle)
        a = np.zeros((n, 14))
        L.orc_world_prims(self.handle, a.ctypes.data_as(C.c_void_p))
        return a

    def camera(self):
        out = (C.c_double * 7)()
        self.oracle.L.orc_world_camera(self.handle, out)
        return list(out[0:3]), list(out[3:6]), out[6]

    def background(self):
        return int(self.oracle.L.orc_world_background(self.handle))

    def probes(self):
        L = self.oracle.L
        return int(L.orc_world_raw_u64_before_last_bvh(self.handle)), int(L.orc_world_next_raw_u64(self.handle))

    def bvh_count(self):
        return int(self.oracle.L.orc_world_bvh_count(self.handle))

    def bvh_info(self, which):
        out = (C.c_int * 3)()
        self.oracle.L.orc_world_bvh_info(self.handle, which, out)
        return {"nodes": out[0], "leaves": out[1], "max_depth": out[2]}

    def bvh_dump(self, which):
        L = self.oracle.L
        n = L.orc_world_bvh_dump(self.handle, which, None, 0)
        a = np.zeros((n, 8))
        L.orc_world_bvh_dump(self.handle, which, a.ctypes.data_as(C.c_void_p), n)
        return a

    def perlin(self, which):
        vec, perm = np.zeros((1024, 3)), np.zeros((3, 1024), dtype=np.int32)
        self.oracle.L.orc_world_perlin(self.handle, which, vec.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p))
        return vec, perm

    def perlin_count(self):
        return int(self.oracle.L.orc_world_perlin_count(self.handle))

    def render(self, camera19, width, height, spp, max_depth=50, seed=None, rng_mode=RNG_PHILOX_SAMPLE, rows=None, threads=0, want_accum=False):
        """Returns (rgb int32 [H][W][3] (only listed rows valid), accum or None, stats dict, seconds)."""
        L = self.oracle.L
        rgb = np.zeros((height, width, 3), dtype=np.int32)
        accum = np.zeros((height, width, 3)) if want_accum else None
        stats = (C.c_uint64 * 10)()
        if rows is None:
            rows_ptr, n_rows = None, 0
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            rows_ptr, n_rows = rows.ctypes.data_as(C.c_void_p), int(rows.size)
        cam = (C.c_double * 19)(*camera19)
        secs = L.orc_render(self.handle, cam, width, height, spp, max_depth, C.c_uint64(self.seed if seed is None else seed), rng_mode, rows_ptr,
                            n_rows, threads, rgb.ctypes.data_as(C.c_void_p), accum.ctypes.data_as(C.c_void_p) if want_accum else None, stats)
        return rgb, accum, dict(zip(STAT_NAMES, [int(x) for x in stats])), float(secs)

    def render_single_light(self, camera19, width, height, spp, light_source, intensity, seed=None, rng_mode=RNG_PHILOX_SAMPLE, rows=None, threads=0, want_accum=False):
        """render() with the Renderer's other RayTracer, SingleLightSourceRayTracer (raytrace.rs:110-135).  Returns (rgb, accum or None)."""
        L = self.oracle.L
        rgb = np.zeros((height, width, 3), dtype=np.int32)
        accum = np.zeros((height, width, 3)) if want_accum else None
        if rows is None:
            rows_ptr, n_rows = None, 0
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            rows_ptr, n_rows = rows.ctypes.data_as(C.c_void_p), int(rows.size)
        L.orc_render_single_light.restype = C.c_double
        L.orc_render_single_light.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_uint64, C.c_int, C.c_void_p, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p]
        L.orc_render_single_light(self.handle, (C.c_double * 19)(*camera19), width, height, spp, (C.c_double * 4)(*(list(light_source) + [intensity])),
                                  C.c_uint64(self.seed if seed is None else seed), rng_mode, rows_ptr, n_rows, threads, rgb.ctypes.data_as(C.c_void_p),
                                  accum.ctypes.data_as(C.c_void_p) if want_accum else None)
        return rgb, accum

    def trace_ray_single_light(self, origin, direction, light_source, intensity, seed, pixel, sample):
        """SingleLightSourceRayTracer::trace of one ray with the Philox stream of (pixel, sample) from its first draw."""
        L = self.oracle.L
        out = (C.c_double * 3)()
        L.orc_trace_ray_single_light.restype = None
        L.orc_trace_ray_single_light.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.orc_trace_ray_single_light(self.handle, (C.c_double * 3)(*origin), (C.c_double * 3)(*direction), (C.c_double * 4)(*(list(light_source) + [intensity])),
                                     C.c_uint64(seed), pixel, sample, out)
        return np.array(list(out))

    def close(self):
        if self.handle:
            self.oracle.L.orc_world_free(self.handle)
            self.handle = None


class OracleScene:
    """The oracle over an arbitrary flattened scene (oracle/oracle_scene.cpp): per-call probes for the directed tests."""

    def __init__(self, oracle, scene_struct):
        self.L = oracle.L
        self.scene = scene_struct  # (keeps the arrays alive)
        self.handle = self.L.orc_scene_new(C.byref(scene_struct))
        if not self.handle:
            raise RuntimeError("oracle cannot build the scene")

    def probe_rays(self, seed, rays, psd):
        rays = np.ascontiguousarray(rays, dtype=np.float64); psd = np.ascontiguousarray(psd, dtype=np.uint32)
        out = np.zeros((len(rays), 24))
        self.L.orc_scene_probe_rays(self.handle, C.c_uint64(seed), len(rays), rays.ctypes.data_as(C.c_void_p), psd.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def render(self, camera19, width, height, spp, max_depth=50, seed=0, rng_mode=RNG_PHILOX_SAMPLE, background=0, top=(0.5, 0.7, 1.0), bottom=(1.0, 1.0, 1.0),
               threads=0, want_accum=False):
        """The whole render loop (raytrace.rs:172-198) over this scene; (rgb int32 [H][W][3], colour sums or None)."""
        rgb = np.zeros((height, width, 3), dtype=np.int32)
        accum = np.zeros((height, width, 3)) if want_accum else None
        self.L.orc_scene_render.restype = C.c_double
        self.L.orc_scene_render(C.c_void_p(self.handle), int(background), (C.c_double * 3)(*top), (C.c_double * 3)(*bottom), (C.c_double * 19)(*camera19), width, height,
                                spp, max_depth, C.c_uint64(seed), rng_mode, None, 0, threads, rgb.ctypes.data_as(C.c_void_p),
                                accum.ctypes.data_as(C.c_void_p) if want_accum else None)
        return rgb, accum

    def trace_rays(self, seed, rays, pixel_sample, max_depth, background=1, top=(0.5, 0.7, 1.0), bottom=(1.0, 1.0, 1.0)):
        """trace_internal (raytrace.rs:79-102) of given rays, each with the Philox stream of its (pixel, sample) behind the four camera draws."""
        rays = np.ascontiguousarray(rays, dtype=np.float64); ps = np.ascontiguousarray(pixel_sample, dtype=np.uint32)
        out = np.zeros((len(rays), 3))
        self.L.orc_scene_trace_rays(C.c_void_p(self.handle), int(background), (C.c_double * 3)(*top), (C.c_double * 3)(*bottom), int(max_depth), C.c_uint64(seed), len(rays),
                                    rays.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def texture_value(self, tex, uvp):
        uvp = np.ascontiguousarray(uvp, dtype=np.float64)
        out = np.zeros((len(uvp), 3))
        self.L.orc_scene_texture_value(self.handle, int(tex), len(uvp), uvp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def turbulence(self, table, points):
        points = np.ascontiguousarray(points, dtype=np.float64)
        out = np.zeros(len(points))
        self.L.orc_scene_turbulence(self.handle, int(table), len(points), points.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def close(self):
        if self.handle:
            self.L.orc_scene_free(self.handle)
            self.handle = None


class Oracle:
    def __init__(self):
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.orc_world_new.restype = C.c_void_p
        L.orc_world_new.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int]
        for f in ("orc_world_free", "orc_world_prim_count", "orc_world_bvh_count", "orc_world_background", "orc_world_perlin_count"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_world_prims.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_world_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_world_next_raw_u64.restype = C.c_uint64
        L.orc_world_next_raw_u64.argtypes = [C.c_void_p]
        L.orc_world_raw_u64_before_last_bvh.restype = C.c_uint64
        L.orc_world_raw_u64_before_last_bvh.argtypes = [C.c_void_p]
        L.orc_world_bvh_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_world_bvh_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_world_perlin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_perlin_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_render.restype = C.c_double
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_camera_new.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + [C.POINTER(C.c_double)]
        L.orc_aabb_hit.argtypes = [C.POINTER(C.c_double)] * 4 + [C.c_double, C.c_double]
        L.orc_sphere_uv.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_to_rgb.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32)]
        L.orc_sphere_hit.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_world_name.restype = C.c_char_p
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_new.argtypes = [C.c_void_p]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_probe_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_texture_value.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_scene_turbulence.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_probe_get_ray.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_probe_to_rgb.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.L = L

    def scene(self, scene_struct):
        return OracleScene(self, scene_struct)

    def get_rays(self, camera19, width, height, seed, ijs):
        ijs = np.ascontiguousarray(ijs, dtype=np.int32)
        out = np.zeros((len(ijs), 6))
        self.L.orc_probe_get_ray((C.c_double * 19)(*camera19), width, height, C.c_uint64(seed), len(ijs), ijs.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def to_rgb_array(self, spp, sums):
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        out = np.zeros(len(sums), dtype=np.int32)
        self.L.orc_probe_to_rgb(spp, len(sums), sums.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def world_names(self):
        return [self.L.orc_world_name(i).decode() for i in range(self.L.orc_world_count())]

    def world(self, name, seed, earth=None):
        if earth is not None:
            earth = np.ascontiguousarray(earth, dtype=np.uint8)
            h = self.L.orc_world_new(name.encode(), seed, earth.ctypes.data_as(C.c_void_p), earth.shape[1], earth.shape[0])
        else:
            h = self.L.orc_world_new(name.encode(), seed, None, 0, 0)
        if not h:
            raise RuntimeError("oracle cannot build world %r" % name)
        return OracleWorld(self, h, name, seed)

    def camera_new(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        out = (C.c_double * 19)()
        d3 = C.c_double * 3
        self.L.orc_camera_new(d3(*lookfrom), d3(*lookat), d3(*vup), vfov, aspect, aperture, focus_dist, out)
        return list(out)

    def pcg64_u64(self, seed, n):
        out = (C.c_uint64 * n)()
        self.L.orc_pcg64_seed_from_u64_stream(C.c_uint64(seed), n, out)
        return [int(x) for x in out]

    def pcg64_new_stream(self, state, stream, n):
        out = (C.c_uint64 * n)()
        m = (1 << 64) - 1
        self.L.orc_pcg64_new_stream(C.c_uint64(state & m), C.c_uint64(state >> 64), C.c_uint64(stream & m), C.c_uint64(stream >> 64), n, out)
        return [int(x) for x in out]

    def pcg64_state(self, seed):
        out = (C.c_uint64 * 4)()
        self.L.orc_pcg64_state_after_seed(C.c_uint64(seed), out)
        return (int(out[1]) << 64) | int(out[0]), (int(out[3]) << 64) | int(out[2])

    def pcg64_uniform(self, seed, lo, hi, n):
        out = (C.c_double * n)()
        self.L.orc_pcg64_uniform(C.c_uint64(seed), C.c_double(lo), C.c_double(hi), n, out)
        return list(out)

    def pcg64_usize(self, seed, lo, hi, n):
        out = (C.c_uint64 * n)()
        self.L.orc_pcg64_usize(C.c_uint64(seed), C.c_uint64(lo), C.c_uint64(hi), n, out)
        return [int(x) for x in out]

    def philox(self, key, ctr):
        out = (C.c_uint32 * 4)()
        self.L.orc_philox4x32_10((C.c_uint32 * 2)(*key), (C.c_uint32 * 4)(*ctr), out)
        return [int(x) for x in out]

    def philox_uniform(self, seed, pixel, sample, n):
        out = (C.c_double * n)()
        self.L.orc_philox_uniform(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), n, out)
        return list(out)

    def aabb_hit(self, mn, mx, o, d, tmin, tmax):
        d3 = C.c_double * 3
        return bool(self.L.orc_aabb_hit(d3(*mn), d3(*mx), d3(*o), d3(*d), tmin, tmax))

    def sphere_uv(self, n):
        out = (C.c_double * 2)()
        self.L.orc_sphere_uv((C.c_double * 3)(*n), out)
        return (out[0], out[1])

    def to_rgb(self, color, spp):
        out = (C.c_int32 * 3)()
        self.L.orc_to_rgb((C.c_double * 3)(*color), spp, out)
        return tuple(out)

    def sphere_hit(self, c, radius, o, d, tmin, tmax):
        out = (C.c_double * 10)()
        d3 = C.c_double * 3
        if not self.L.orc_sphere_hit(d3(*c), radius, d3(*o), d3(*d), tmin, tmax, out):
            return None
        return {"t": out[0], "p": list(out[1:4]), "normal": list(out[4:7]), "u": out[7], "v": out[8], "front_face": bool(out[9])}
