"""Directed device tests: the device functions of every precision build, evaluated on chosen inputs through the probes of the
debug library (mlrt_debug_probe_*, build.py --debug), against the committed golden vectors the ORACLE produced
(tests/golden/directed_vectors.npz; tests/directed_cases.py lists what each group probes and the reference lines behind it).

The probes walk the REAL packed scene of a context (SAH / wrappers / payload slots and all) with the real step arithmetic; only
the megakernel's lane scheduling is replaced by a loop.  Tolerances, per function:
  * f64 builds (fp64 product over f32 culling entries, cull_f64, strict): hit / scattered / front-face / material / draw-index
    flags EQUAL; t, p, normal, scattered ray, attenuation within 1e-12 relative (device libm vs glibc last-ulp differences in
    sin / cos / cbrt / acos / atan2 / log; the arithmetic itself is the reference's operation order)
  * fp32 fast build: the same flags equal except where an f32 rounding sits within 1e-5 of a decision boundary (listed below);
    t within 1e-6 relative (the sphere quadratic is f64 with a Newton square root), normals / uv / directions / colours 2e-5 abs
  * culling test: NEVER a miss where the reference's AABB::hit hits (all builds); the strict build equal to it; the other builds
    may pass a miss only within their documented slack of the box (1e-6 relative of the scene scale)
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import directed_cases as dc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "directed_vectors.npz"))
BUILDS = ["fp32", "fp64", "fp64_cull64", "fp64_strict"]


@pytest.fixture(scope="module")
def dbg():
    """The debug library in THIS process (the probes only exist there); tests in this module never touch the release library."""
    root = os.path.dirname(HERE)
    lib = os.path.join(root, "mu-lambda-raytracer_amd", "lib", "libmlrt_hip_debug.so")
    if not os.path.exists(lib):
        subprocess.run([sys.executable, os.path.join(root, "mu-lambda-raytracer_amd", "build.py"), "--debug"], check=True, capture_output=True)
    try:
        import torch  # noqa: F401  (its HIP runtime must be the one loaded, see conftest.py)
    except Exception:
        pass
    spec = importlib.util.spec_from_file_location("mlrt_debug_pkg", os.path.join(root, "mu-lambda-raytracer_amd", "__init__.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_PATH = lib
    L = m.lib()
    L.mlrt_debug_probe_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_camera.argtypes = [C.c_void_p, C.POINTER(m.CameraStruct), C.POINTER(m.Params), C.c_int32, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_slab.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_turbulence.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_texture.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_to_rgb.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.mlrt_debug_probe_divide.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    return m


def context(m, scene, build):
    precision = {"fp32": m.PRECISION_FP32, "fp64": m.PRECISION_FP64, "fp64_cull64": m.PRECISION_FP64, "fp64_strict": m.PRECISION_FP64_STRICT}[build]
    opts = m.options(force_full=1, cull_f64=1 if build == "fp64_cull64" else 0)
    ctx = C.c_void_p()
    rc = m.lib().mlrt_context_create_opts(C.byref(scene), 0, precision, C.byref(opts), C.byref(ctx))
    assert rc == 0, m.lib().mlrt_last_error()
    return ctx


def check(m, rc):
    assert rc == 0, m.lib().mlrt_last_error()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("name", sorted(dc.SCENES))
def test_one_bounce_of_chosen_rays(dbg, build, name):
    m = dbg
    sc = dc.SCENES[name](m)
    ctx = context(m, sc.to_ctypes(), build)
    rays, psd, expect, labels = GOLD["rays_" + name], GOLD["psd_" + name], GOLD["expect_" + name], [str(x) for x in GOLD["labels_" + name]]
    got = np.zeros((len(rays), 24))
    check(m, m.lib().mlrt_debug_probe_rays(ctx, dc.SEED, len(rays), ptr(rays), ptr(psd), ptr(got)))
    m.lib().mlrt_context_destroy(ctx)
    f64 = build != "fp32"
    bad = []
    for i, label in enumerate(labels):
        g, e = got[i], expect[i]
        if label.startswith("inplane:"):
            # a ray exactly in the plane of an AARect: the reference reports a hit at t = NaN (0/0 passes every comparison of
            # aarects.rs:47-54, quirk Q16).  No parity is claimed there; the device must simply come back with a verdict.
            assert g[0] in (0.0, 1.0)
            continue
        if label.startswith("boxplane:") and build != "fp64_strict":
            # a ray exactly in a face plane of a primitive's bounding box: the reference's AABB::hit turns 0/0 into a miss, so the
            # answer depends on which boxes its BVH tests (quirk Q17); only the strict build shares that topology
            assert g[0] in (0.0, 1.0)
            continue
        flags = [0, 10, 11, 21, 22]  # hit, front_face, scattered, draw index, material
        if g[0] != e[0]:
            if not f64 and "just outside the edge" in label:
                continue  # 2^-30 outside a rect's edge: below the f32 resolution of the fast build's rect test
            bad.append((label, "hit", g[0], e[0]))
            continue
        if e[0] == 0:
            if g[21] != e[21]:
                bad.append((label, "draws", g[21], e[21]))
            continue
        for k in flags:
            if g[k] != e[k]:
                # fp32: an absorbed / scattered verdict of a fuzzed reflection within f32 rounding of the surface, and a refraction
                # draw against a reflectance within 1e-6 of the word, may flip; nothing of the kind may happen in an f64 build
                bad.append((label, "flag %d" % k, g[k], e[k]))
        rel_t = 1e-12 if f64 else 2e-6
        tol = 1e-12 if f64 else 3e-5
        if abs(g[1] - e[1]) > rel_t * max(1.0, abs(e[1])):
            bad.append((label, "t", g[1], e[1]))
        scale = max(1.0, float(np.abs(e[2:5]).max()))
        if np.abs(g[2:5] - e[2:5]).max() > (1e-12 if f64 else 4e-6) * scale:
            bad.append((label, "p", g[2:5], e[2:5]))
        if np.abs(g[5:8] - e[5:8]).max() > tol:
            bad.append((label, "normal", g[5:8], e[5:8]))
        # u, v: the reference computes sphere_uv for every accepted sphere intersection (quirk Q11) though only Image textures read
        # it; the device computes it where the hit material's texture chain samples an image, and leaves 0 otherwise
        if (label.startswith("sphere_uv") or g[8] != 0.0 or g[9] != 0.0) and np.abs(g[8:10] - e[8:10]).max() > tol:
            bad.append((label, "uv", g[8:10], e[8:10]))
        if g[11] == e[11]:
            if e[11] == 1 and np.abs(g[12:15] - e[12:15]).max() > (1e-12 if f64 else 4e-6) * scale:
                bad.append((label, "scattered origin", g[12:15], e[12:15]))
            if e[11] == 1 and np.abs(g[15:18] - e[15:18]).max() > tol * max(1.0, float(np.abs(e[15:18]).max())):
                bad.append((label, "scattered direction", g[15:18], e[15:18]))
            if np.abs(g[18:21] - e[18:21]).max() > tol * max(1.0, float(np.abs(e[18:21]).max())):
                bad.append((label, "attenuation/emitted", g[18:21], e[18:21]))
    assert not bad, "%d of %d rays differ; first: %s" % (len(bad), len(labels), bad[:6])


@pytest.mark.parametrize("build", BUILDS)
def test_culling_test_never_rejects_what_the_reference_box_test_accepts(dbg, build):
    m = dbg
    sc = dc.base(m)
    sc.add_prim(sc.box((0, 0, 0), (1, 1, 1), sc.lambertian(sc.t_white)))
    ctx = context(m, sc.to_ctypes(), build)
    boxes, rays, tmax, labels = GOLD["aabb_boxes"], GOLD["aabb_rays"], GOLD["aabb_tmax"], [str(x) for x in GOLD["aabb_labels"]]
    hit = np.zeros(len(boxes), dtype=np.int32)
    check(m, m.lib().mlrt_debug_probe_slab(ctx, len(boxes), ptr(boxes), ptr(rays), ptr(tmax), ptr(hit)))
    m.lib().mlrt_context_destroy(ctx)
    expect = GOLD["expect_aabb"]
    # the reference's five unit tests x two boxes: every build answers as the reference does
    assert list(hit[:10]) == [1, 1, 0, 1, 0] * 2, build
    missed = [labels[i] for i in range(len(hit)) if expect[i] == 1 and hit[i] == 0]
    assert not missed, (build, missed[:5])  # conservative: never a miss where AABB::hit hits
    extra = [labels[i] for i in range(len(hit)) if expect[i] == 0 and hit[i] == 1]
    if build == "fp64_strict":
        assert not extra, extra[:5]  # the reference's own arithmetic: equal
    else:
        # a culling test may pass a ray the exact test rejects only within its slack: grazing rays (the "silhouette" group aims
        # within 1e-9 .. 1e-4 of the boxes' faces), closest-so-far exactly at the entry face, and the zero-thickness box (padded)
        # (and rays lying exactly in a face plane: 0/0 on that axis, which the reference's NaN-ignoring min / max turn into a miss)
        assert all(l.startswith("silhouette") or "exactly at the entry face" in l or "zero thickness" in l or "corner, almost" in l or "along the" in l for l in extra), extra[:5]
        assert len(extra) <= (60 if build == "fp32" or build == "fp64" else 25), (build, len(extra))


@pytest.mark.parametrize("build", BUILDS)
def test_textures(dbg, build):
    m = dbg
    sc = dc.scene_primitives(m)
    ctx = context(m, sc.to_ctypes(), build)
    f64 = build != "fp32"
    uv, pts = GOLD["tex_uv"], GOLD["tex_pts"]
    out = np.zeros((len(uv), 3))
    check(m, m.lib().mlrt_debug_probe_texture(ctx, sc.t_image, len(uv), ptr(uv), ptr(out)))
    # image_texture.rs:16-28: nearest texel, u = 1 and v = 0 clamp to the last column / row: bytes / 255, exact in f64, 1 ulp in f32
    assert np.abs(out - GOLD["expect_tex_image"]).max() <= (0.0 if f64 else 1e-7)
    for tex, key in ((sc.t_noise, "expect_tex_noise"), (sc.t_checker, "expect_tex_checker")):
        out = np.zeros((len(pts), 3))
        check(m, m.lib().mlrt_debug_probe_texture(ctx, tex, len(pts), ptr(pts), ptr(out)))
        d = np.abs(out - GOLD[key])
        if f64:
            assert d.max() < 1e-12, (key, d.max())
        else:
            # f32 lattice arithmetic: |p| up to 1e4 loses 1e-3 of a cell in the 64x octave; compare where the point is f32-benign,
            # and a checker may pick its other branch when a sine sits at a zero crossing
            benign = np.abs(pts[:, 2:5]).max(axis=1) < 400.0
            frac = float(np.mean(d[benign].max(axis=1) > 5e-3))
            assert frac <= 0.1, (key, frac)
    turb_pts = GOLD["turb_pts"]
    serial, wave = np.zeros(len(turb_pts)), np.zeros(len(turb_pts))
    check(m, m.lib().mlrt_debug_probe_turbulence(ctx, 0, len(turb_pts), ptr(turb_pts), ptr(serial), ptr(wave)))
    m.lib().mlrt_context_destroy(ctx)
    e = GOLD["expect_turbulence"]
    if f64:
        assert np.abs(serial - e).max() < 1e-13  # textures.rs:76-134, negative lattice coordinates included
    else:
        small = np.abs(turb_pts).max(axis=1) < 40.0
        assert np.abs(serial - e)[small].max() < 2e-4 and np.abs(wave - e)[small].max() < 2e-4
        assert np.abs(wave - serial)[small].max() < 1e-5  # the wave-level evaluation of the shading batch: same terms, other order


@pytest.mark.parametrize("build", BUILDS)
def test_camera_rays(dbg, build):
    m = dbg
    sc = dc.scene_primitives(m)
    ctx = context(m, sc.to_ctypes(), build)
    cams, ijs, w, h = dc.camera_cases(m)
    world = m.World("simple", dc.SEED)
    for k, cam in enumerate(cams):
        params = m.make_params(world, w, h, 1, seed=dc.SEED)
        out = np.zeros((len(ijs), 6))
        check(m, m.lib().mlrt_debug_probe_camera(ctx, C.byref(cam), C.byref(params), len(ijs), ptr(ijs), ptr(out)))
        e = GOLD["expect_cam%d" % k]
        if build != "fp32":
            # raytrace.rs:191-193 + camera.rs:40-48 in the reference's operation order; the lens sample's sin / cos are libm
            assert np.abs(out - e).max() <= 1e-12 * np.abs(e).max(), (k, np.abs(out - e).max())
            if k == 0:
                assert np.array_equal(out, e)  # pinhole: no transcendental at all
        else:
            # fast build: sub-pixel position to 2^-13 pixel, direction in f32, hardware sin / cos for the lens
            assert np.array_equal(out[:, :3], e[:, :3]) if k == 0 else np.abs(out[:, :3] - e[:, :3]).max() < 2e-5
            assert np.abs(out[:, 3:] - e[:, 3:]).max() < 3e-5 * np.abs(e[:, 3:]).max()
    world.close()
    m.lib().mlrt_context_destroy(ctx)


@pytest.mark.parametrize("build", BUILDS)
def test_to_rgb_of_one_accumulated_sample(dbg, build):
    """to_fixed (one finished sample into the 2^-32 fixed-point sum) + the resolve kernel's to_rgb against raytrace.rs:59-68.
    Where they differ BY DESIGN: a NaN sample contributes 0 to the sum (the reference's sum would be NaN and the pixel 0 - same
    byte for a one-sample pixel), negatives clamp to 0 (as the reference's clamp does after the square root of a negative mean:
    NaN -> 0), and a single sample saturates at 2^20 (far above the 1.0 that already renders as 255)."""
    m = dbg
    sc = dc.scene_primitives(m)
    ctx = context(m, sc.to_ctypes(), build)
    vals = GOLD["to_rgb_vals"]
    for spp in (1, 4):
        out = np.zeros(len(vals), dtype=np.int32)
        check(m, m.lib().mlrt_debug_probe_to_rgb(ctx, spp, len(vals), ptr(vals), ptr(out)))
        e = GOLD["expect_to_rgb_spp%d" % spp]
        if build == "fp32":
            # the f32 fixed-point conversion truncates below 2^-32 and rounds the sample to f32 first: one level at most, never at 0 / 255
            assert np.abs(out - e).max() <= 1 and np.array_equal(out[e == 0], e[e == 0]) and np.array_equal(out[e == 255], e[e == 255])
        else:
            assert np.array_equal(out, e), (spp, out, e)
    m.lib().mlrt_context_destroy(ctx)


@pytest.mark.parametrize("build", ["fp64", "fp64_cull64", "fp64_strict"])
def test_quotients_over_a_shared_divisor_are_the_divisions(dbg, build):
    """Divisor::over (mlrt_kernel.hip: the reciprocal refinement of x / a done once per divisor, the rest per numerator) against
    x / a itself, bit for bit, on 2.6e8 operand pairs made on the device: inside its domain (divisors in 2^-300 .. 2^300, numerators
    in 2^-600 .. 2^600 or zero, either sign) and with divisors OUTSIDE it (down to denormal-adjacent and up to the largest
    exponents), which must take the division itself.  The strict build only ever divides: trivially equal."""
    m = dbg
    sc = dc.base(m)
    sc.add_prim(sc.box((0, 0, 0), (1, 1, 1), sc.lambertian(sc.t_white)))
    ctx = context(m, sc.to_ctypes(), build)
    out = (C.c_ulonglong * 4)()
    for seed, (a_lo, a_hi, x_lo, x_hi) in enumerate([(-300, 299, -600, 599), (-40, 40, -80, 80), (-3, 3, -8, 8), (-1022, 1023, -300, 300), (-300, 299, -300, 300)]):
        check(m, m.lib().mlrt_debug_probe_divide(ctx, 256, 160, 1000 + seed, a_lo, a_hi, x_lo, x_hi, out))
        assert out[0] == 0, (build, (a_lo, a_hi, x_lo, x_hi), int(out[0]), [hex(int(v)) for v in out[1:]])
    m.lib().mlrt_context_destroy(ctx)


@pytest.mark.parametrize("build", ["fp64", "fp64_strict"])
def test_the_mediums_logarithm_over_all_its_inputs(dbg, build):
    """log_uniform (mlrt_kernel.hip; ConstantMedium's ln of a uniform, volumes.rs:48) on EVERY input it can get - the 2^24 uniforms
    k 2^-24 - against glibc's log (what the oracle, and Rust's f64::ln, call): never more than one unit in the last place away, the
    same double for nine inputs in ten, and -inf at 0 (no hit, as with ln).  The frames of the worlds with media are held to the
    oracle's with the tolerance this leaves (test_gpu_parity.py)."""
    m = dbg
    sc = dc.base(m)
    sc.add_prim(sc.box((0, 0, 0), (1, 1, 1), sc.lambertian(sc.t_white)))
    ctx = context(m, sc.to_ctypes(), build)
    L = m.lib()
    L.mlrt_debug_probe_log.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    differ, worst, chunk = 0, 0.0, 1 << 22
    for k0 in range(0, 1 << 24, chunk):
        out = np.zeros(chunk)
        check(m, L.mlrt_debug_probe_log(ctx, k0, chunk, ptr(out)))
        x = np.arange(k0, k0 + chunk, dtype=np.float64) * 2.0 ** -24
        if k0 == 0:
            assert out[0] == -np.inf
            out, x = out[1:], x[1:]
        want = np.log(x)
        bad = out != want
        differ += int(bad.sum())
        if bad.any():
            worst = max(worst, float((np.abs(out[bad] - want[bad]) / np.spacing(np.abs(want[bad]))).max()))
    assert worst <= 1.0 and differ < 0.10 * (1 << 24), (differ, worst)
    m.lib().mlrt_context_destroy(ctx)
