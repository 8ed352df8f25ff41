"""Chosen rays through the RELEASE render kernel - tile queue, lane state machine, special and shading batches, regeneration,
fixed-point accumulation and all - not the probe loop of the debug library (tests/test_gpu_directed.py).

A "ray frame" (test hook mlrt_debug_ray_frame, mlrt_api.cpp) renders a W x H frame in which sample s of pixel p starts on a ray
from a table instead of the camera's; everything behind the primary ray is the product path.  The f64 colour sum of every pixel
is compared with the ORACLE's RecursiveRayTracer::trace (/root/reference/src/raytrace.rs:79-107, oracle_scene.cpp
orc_scene_trace_rays) of the same rays, each with the Philox stream of its (pixel, sample) positioned behind the four camera draws.

Rays: the directed sets of tests/directed_cases.py (axis-parallel rays, origins inside boxes and spheres, tangent and behind-origin
spheres, closed rect edges, media entered from inside, total internal reflection, transforms ...) on its three scenes, three bounces
deep, three samples (three different streams) per ray; and random rays on sphere-only scenes, which exercise the pair walk's kernel
variant.  Every precision build, with the DEFAULT options: each scene runs on the kernel variant the product would pick for it.

Tolerances.  A finished sample is rounded to 2^-32 when it is accumulated (to_fixed), so an f64 pixel may differ from the oracle's
sum by spp x 2^-33, plus the last-ulp differences between the device's libm and glibc in worlds that call sin / log / acos / atan2
(1e-11 relative).  Rays lying exactly in a rect's plane (quirk Q16) or in a face plane of a bounding box (Q17) carry no parity
except - Q17 - for the strict build, as in the directed tests.  Fast build: 1e-3 absolute on 97 % of the pixels (grazing rays may
flip a decision in f32, as documented for that mode).

Last: the documented NaN divergence (include/mlrt.h): a sample whose radiance is NaN - reachable through Q16 rays only -
contributes 0 on the device, where the reference's sum becomes NaN and the pixel 0."""
import ctypes as C

import numpy as np
import pytest

import directed_cases as dc
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu
BUILDS = ["fp64", "fp64_cull64", "fp64_strict", "fp32"]
TOP, BOTTOM = (0.5, 0.7, 1.0), (1.0, 1.0, 1.0)


def make_context(m, scene, build):
    precision = {"fp32": m.PRECISION_FP32, "fp64": m.PRECISION_FP64, "fp64_cull64": m.PRECISION_FP64, "fp64_strict": m.PRECISION_FP64_STRICT}[build]
    opts = m.options(cull_f64=1 if build == "fp64_cull64" else 0)
    ctx = C.c_void_p()
    rc = m.lib().mlrt_context_create_opts(C.byref(scene), 0, precision, C.byref(opts), C.byref(ctx))
    assert rc == 0, m.lib().mlrt_last_error()
    return ctx


def ray_frame(m, ctx, rays, spp, max_depth, seed):
    """rays [n][6]: pixel i of row 0 (and of row 1: a frame has at least two rows) starts all its samples on rays[i].  Returns the
    colour sums [2][W][3], the table [2 * W * spp][6] and its (pixel, sample) ids."""
    n = len(rays)
    width = max(2, n)
    table = np.zeros((2, width, spp, 6))
    for i in range(width):
        table[:, i, :, :] = rays[i % n]
    table = np.ascontiguousarray(table.reshape(-1, 6))
    p = m.Params()
    p.image_width, p.image_height, p.samples_per_pixel, p.max_depth, p.seed = width, 2, spp, max_depth, seed
    p.background_kind = m.BG_GRADIENT
    p.background_top[:] = TOP
    p.background_bottom[:] = BOTTOM
    accum = np.zeros((2, width, 3))
    L = m.lib()
    L.mlrt_debug_ray_frame.argtypes = [C.c_void_p, C.POINTER(m.Params), C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.mlrt_debug_ray_frame(ctx, C.byref(p), table.ctypes.data_as(C.c_void_p), accum.ctypes.data_as(C.c_void_p), None)
    assert rc == 0, L.mlrt_last_error()
    ids = np.zeros((2 * width * spp, 2), dtype=np.uint32)
    ids[:, 0] = np.repeat(np.arange(2 * width), spp)
    ids[:, 1] = np.tile(np.arange(spp), 2 * width)
    return accum, table, ids


def compare(build, accum, want, labels, spp):
    f64 = build != "fp32"
    n = len(labels)
    bad, checked = [], 0
    for j in range(2):
        for i in range(accum.shape[1]):
            label = labels[i % n]
            if label.startswith("inplane:") or (label.startswith("boxplane:") and build != "fp64_strict"):
                continue  # quirks Q16 / Q17: no parity claimed (see the module docstring)
            checked += 1
            g, e = accum[j, i], want[j, i]
            if not np.all(np.isfinite(e)):
                continue  # (a NaN radiance: covered by the test at the end of this file)
            tol = (spp * 2.0 ** -32 + 1e-11 * max(1.0, float(np.abs(e).max()))) if f64 else 1e-3 * max(1.0, float(np.abs(e).max()))
            if np.abs(g - e).max() > tol:
                bad.append((label, j, i, g.tolist(), e.tolist()))
    allowed = 0 if f64 else int(0.03 * checked)
    assert len(bad) <= allowed, "%d of %d pixels differ (allowed %d); first: %s" % (len(bad), checked, allowed, bad[:4])


@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("name", sorted(dc.SCENES))
def test_directed_rays_through_the_release_kernel(mlrt, oracle, build, name):
    m = mlrt
    sc = dc.SCENES[name](m)
    scene = sc.to_ctypes()
    rays, labels = dc.rays_for(name)
    rays = np.asarray(rays, dtype=np.float64)
    spp, max_depth = 3, 3
    ctx = make_context(m, scene, build)
    accum, table, ids = ray_frame(m, ctx, rays, spp, max_depth, dc.SEED)
    m.lib().mlrt_context_destroy(ctx)
    assert accum.any()
    osc = OracleScene(oracle, scene)
    want = osc.trace_rays(dc.SEED, table, ids, max_depth, background=1, top=TOP, bottom=BOTTOM).reshape(2, -1, spp, 3).sum(axis=2)
    osc.close()
    compare(build, accum, want, labels, spp)


def sphere_scene(m, rng, n):
    sc = dc.MiniScene(m)
    mats = [sc.lambertian(sc.solid(*rng.uniform(0.2, 0.9, 3))) for _ in range(4)] + [sc.metal(tuple(rng.uniform(0.4, 1.0, 3)), float(rng.uniform(0.0, 0.6))), sc.dielectric(1.5)]
    prims = [sc.sphere(tuple(rng.uniform(-12, 12, 3) + np.array([0, 0, -20.0])), float(rng.uniform(0.3, 2.5)), mats[int(rng.integers(len(mats)))]) for _ in range(n)]
    sc.add_prim(sc.sphere((0.0, -1012.0, -20.0), 1000.0, mats[0]))  # a ground sphere, as in `random` (worlds.rs:83)
    cut = n // 3
    for p in prims[:cut]:
        sc.add_prim(p)
    sc.add_bvh(prims[cut:])
    return sc


@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("n_spheres", [5, 300])
def test_random_rays_through_the_sphere_only_variants(mlrt, oracle, build, n_spheres):
    """Sphere-only scenes run on the pair walk (f32 culling builds) or the threaded list (cull_f64, strict): deep paths (max_depth 12)
    from random origins - outside, among and inside the spheres - against the oracle's trace; 1500 rays x 2 samples."""
    m = mlrt
    rng = np.random.default_rng(77 + n_spheres)
    sc = sphere_scene(m, rng, n_spheres)
    scene = sc.to_ctypes()
    n = 1500
    origin = np.where((np.arange(n) % 4 == 0)[:, None], rng.uniform(-12, 12, (n, 3)) + np.array([0, 0, -20.0]), rng.uniform(-40, 40, (n, 3)))
    target = rng.uniform(-12, 12, (n, 3)) + np.array([0, 0, -20.0])
    d = target - origin
    d *= rng.uniform(0.1, 20.0, (n, 1)) / np.linalg.norm(d, axis=1)[:, None]
    d[::13, 1] = 0.0
    rays = np.concatenate([origin, d], axis=1)
    spp, max_depth = 2, 12
    ctx = make_context(m, scene, build)
    want_variant = 2 if build in ("fp64", "fp32") else 1  # the pair structure where the build has it
    assert m.lib().mlrt_context_kernel_variant(ctx) == want_variant
    accum, table, ids = ray_frame(m, ctx, rays, spp, max_depth, 4242)
    m.lib().mlrt_context_destroy(ctx)
    osc = OracleScene(oracle, scene)
    want = osc.trace_rays(4242, table, ids, max_depth, background=1, top=TOP, bottom=BOTTOM).reshape(2, -1, spp, 3).sum(axis=2)
    osc.close()
    compare(build, accum, want, ["random"] * n, spp)


def test_a_nan_sample_contributes_zero_where_the_reference_zeroes_the_pixel(mlrt, oracle):
    """Quirk Q16: a ray lying exactly in the plane of an AARect gets t = 0/0 = NaN from aarects.rs:46, which passes every comparison of
    :47-54 - a hit at t = NaN, a NaN hit point, NaN radiance.  The reference adds it to the pixel's sum (raytrace.rs:188-197), the sum is
    NaN and to_rgb (raytrace.rs:59-68) turns the pixel into (0, 0, 0).  The device's accumulation is 2^-32 fixed point and cannot hold a
    NaN: such a sample contributes 0 (mlrt_kernel.hip to_fixed) and the pixel keeps the other samples' light.  Documented divergence
    (DESIGN.md section 5), reachable only by measure-zero rays; asserted here so that it stays what the documents say."""
    m = mlrt
    sc = dc.base(m)
    marble = sc.lambertian(sc.t_noise)  # a texture that looks at the hit point: NaN in, NaN out (textures.rs:163-167) - a solid colour would not care
    sc.add_prim(sc.rect(0, 1, -1.0, 1.0, -1.0, 1.0, -5.0, marble))  # an XY rect in the plane z = -5
    sc.add_prim(sc.sphere((40.0, 30.0, -5.0), 1.0, sc.lambertian(sc.t_white)))  # (off the ray's line: a later hit at a finite t would replace the NaN one)
    scene = sc.to_ctypes()
    in_plane = [-3.0, 0.25, -5.0, 1.0, 0.0, 0.0]   # origin in the rect's plane, direction in the plane: (k - o.z) / d.z = 0 / 0
    sky = [0.0, 0.0, 0.0, 0.0, 1.0, 0.0]           # misses everything: the gradient background at unit y
    osc = OracleScene(oracle, scene)
    ids = np.array([[0, 0], [0, 1]], dtype=np.uint32)
    ref = osc.trace_rays(9, np.array([in_plane, sky]), ids, 4, background=1, top=TOP, bottom=BOTTOM)
    osc.close()
    assert np.isnan(ref[0]).any(), "the reference's radiance for the in-plane ray is NaN (quirk Q16)"
    assert np.allclose(ref[1], TOP)
    for build in ("fp64", "fp64_strict"):
        ctx = make_context(m, scene, build)
        # pixel 0: sample 0 = the NaN ray, sample 1 = the sky ray (a table of per-sample rays: [pixel][sample])
        width, spp = 2, 2
        table = np.zeros((2, width, spp, 6))
        table[:, :, :, :] = sky
        table[0, 0, 0, :] = in_plane
        p = m.Params()
        p.image_width, p.image_height, p.samples_per_pixel, p.max_depth, p.seed = width, 2, spp, 4, 9
        p.background_kind = m.BG_GRADIENT
        p.background_top[:] = TOP
        p.background_bottom[:] = BOTTOM
        accum = np.zeros((2, width, 3))
        rgb = np.zeros((2, width, 3), dtype=np.int32)
        L = m.lib()
        L.mlrt_debug_ray_frame.argtypes = [C.c_void_p, C.POINTER(m.Params), C.c_void_p, C.c_void_p, C.c_void_p]
        flat = np.ascontiguousarray(table.reshape(-1, 6))
        assert L.mlrt_debug_ray_frame(ctx, C.byref(p), flat.ctypes.data_as(C.c_void_p), accum.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p)) == 0, L.mlrt_last_error()
        L.mlrt_context_destroy(ctx)
        # the device: NaN sample -> 0, so the pixel holds the sky sample alone; the reference would hold NaN -> (0, 0, 0)
        assert np.allclose(accum[0, 0], TOP, atol=1e-9), (build, accum[0, 0])
        assert np.allclose(accum[0, 1], 2 * np.array(TOP), atol=1e-9)
        assert rgb[0, 0].min() > 0


def test_a_rect_that_is_a_bvh_leaf(mlrt, oracle):
    """No reference world puts an AARect into a BHV; if one did, AARect::bounding_box (aarects.rs:66-76) - maximum set to (a0_v0, a1_v0):
    a point - would make the inner node above it enclose one corner of the rect only, and the reference would cull most rays that hit
    the rect.  The product modes (FP64, FP32) rebuild their BVHs from the primitives' true extents and render the intended geometry:
    the same picture as the rect as a plain list member.  The strict build walks the caller's bvh_nodes with the caller's bounds and so
    agrees with the reference on the scene as given (include/mlrt.h, mlrt_rect).  Pinned here both ways."""
    m = mlrt

    def scene(in_bvh):
        sc = dc.MiniScene(m)
        red, grey = sc.lambertian(sc.solid(0.65, 0.05, 0.05)), sc.lambertian(sc.solid(0.5, 0.5, 0.5))
        rect = sc.rect(0, 1, -2.0, 2.0, -2.0, 2.0, -5.0, red)
        ball = sc.sphere((-3.0, -3.0, -5.0), 0.5, grey)  # next to the corner the reference takes for the whole rect
        far_box = sc.box((20.0, 20.0, -8.0), (21.0, 21.0, -7.0), grey)  # (keeps the scene on the FULL variants)
        if in_bvh:
            sc.add_bvh([rect, ball])
        else:
            sc.add_prim(rect); sc.add_prim(ball)
        sc.add_prim(far_box)
        return sc.to_ctypes()

    rng = np.random.default_rng(5)
    n = 400
    target = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), np.full(n, -5.0)], axis=1)  # points of the rect
    origin = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-1.0, 1.0, n), np.full(n, 3.0)], axis=1)
    rays = np.concatenate([origin, target - origin], axis=1)
    tree, flat = scene(True), scene(False)
    osc = OracleScene(oracle, tree)
    probe = osc.probe_rays(1, rays, np.tile(np.array([[0, 0, 4]], dtype=np.uint32), (n, 1)))
    osc.close()
    assert (probe[:, 0] == 0).mean() > 0.5, "the reference culls most of these rays (its inner node encloses a corner of the rect only)"
    spp, depth = 2, 3
    got = {}
    for build in ("fp64", "fp32", "fp64_strict"):
        ctx = make_context(m, tree, build)
        got[build], table, ids = ray_frame(m, ctx, rays, spp, depth, 1)
        m.lib().mlrt_context_destroy(ctx)
    osc_tree, osc_flat = OracleScene(oracle, tree), OracleScene(oracle, flat)
    want_tree = osc_tree.trace_rays(1, table, ids, depth, background=1, top=TOP, bottom=BOTTOM).reshape(2, -1, spp, 3).sum(axis=2)
    want_flat = osc_flat.trace_rays(1, table, ids, depth, background=1, top=TOP, bottom=BOTTOM).reshape(2, -1, spp, 3).sum(axis=2)
    osc_tree.close(); osc_flat.close()
    assert np.abs(want_tree - want_flat).max() > 0.1  # (the two readings of the scene do differ)
    compare("fp64", got["fp64"], want_flat, ["rect in a BVH"] * n, spp)        # product: the intended geometry
    compare("fp32", got["fp32"], want_flat, ["rect in a BVH"] * n, spp)
    compare("fp64_strict", got["fp64_strict"], want_tree, ["rect in a BVH"] * n, spp)  # strict: the reference on the scene as given
