"""Randomised device parity: seeded random scenes (every primitive, material, texture, transform and medium kind the C ABI
accepts, BVHs of arbitrary topology) and random rays, one bounce each - closest hit over the REAL packed scene (SAH, wrappers,
payload slots), hit attributes, material scatter, medium draws - of every precision build against the ORACLE evaluated on the
same flattened scene at run time (oracle/oracle_scene.cpp: hittable.rs:53-69, bhv.rs:147-165, shapes.rs, aarects.rs,
transforms.rs, volumes.rs:26-65, materials.rs).  tests/test_gpu_directed.py holds the chosen edge cases; this file is the bulk:
random inputs sit on a decision boundary with probability ~0, so the f64 builds must agree with the oracle on every flag of every ray.

Tolerances as in test_gpu_directed.py: f64 builds flags equal, values 1e-12 relative (device libm vs glibc); fast build: the
closest hit may differ where two surfaces are within f32 rounding of each other or a ray grazes one (< 0.3 % of the rays), the
rest within 3e-5."""
import numpy as np
import pytest

import directed_cases as dc
from oracle_binding import OracleScene
from test_gpu_directed import BUILDS, check, context, dbg, ptr  # noqa: F401  (dbg: the debug-library fixture)

pytestmark = pytest.mark.gpu
N_RAYS = 3000


def random_scene(m, seed, size=1.0, shift=(0.0, 0.0, 0.0)):
    """(scene, lo, hi): objects inside the box [lo, hi].  `size`, `shift`: the same scene at another scale / far from the origin."""
    rng = np.random.default_rng(seed)
    shift = np.asarray(shift, dtype=np.float64)
    sc = dc.base(m)
    u = rng.uniform
    mats = [sc.lambertian(sc.t_white), sc.lambertian(sc.t_image), sc.lambertian(sc.t_checker), sc.lambertian(sc.t_noise), sc.light(sc.t_light)]
    mats += [sc.metal(tuple(u(0.3, 1.0, 3)), float(u(0.0, 1.0))) for _ in range(3)]
    mats += [sc.metal((0.9, 0.8, 0.7), 0.0), sc.dielectric(1.5), sc.dielectric(float(u(1.05, 2.4)))]
    isos = [sc.isotropic(sc.solid(*u(0.1, 1.0, 3))), sc.isotropic(sc.t_checker)]
    lo, hi = size * np.array([-12.0, -8.0, -30.0]) + shift, size * np.array([12.0, 8.0, -6.0]) + shift
    pick = lambda: mats[int(rng.integers(len(mats)))]  # noqa: E731
    point = lambda: u(lo, hi)  # noqa: E731

    def transform():
        """(index, world point -> object-space point): Translate(offset, Rotate(axis, angle, .)) as transforms.rs:31-42, 106-124."""
        k = int(rng.integers(4))
        if k == 0:
            return -1, lambda w: w
        axis = int(rng.integers(3)) if k >= 2 else None
        off = size * u(-3.0, 3.0, 3) if k != 2 else np.zeros(3)
        t = sc.transform(rotate_axis=axis, degrees=float(u(-80.0, 80.0)), offset=tuple(off) if k != 2 else None)
        sin_t, cos_t = sc.transforms[t]["sin_theta"], sc.transforms[t]["cos_theta"]

        def to_object(w):
            v = np.asarray(w, dtype=np.float64) - off
            if axis is not None:  # rotate_back
                a0, a2 = (axis + 2) % 3, (axis + 1) % 3
                v0, v2 = v[a0], v[a2]
                v[a0], v[a2] = cos_t * v0 - sin_t * v2, sin_t * v0 + cos_t * v2
            return v
        return t, to_object

    def sphere(place, rmax=2.5):
        return sc.sphere(tuple(place(point())), size * float(u(0.2, rmax)), pick())

    def box(place, smax=3.0):
        c = place(point())
        half = size * u(0.15, smax / 2, 3)
        return sc.box(tuple(c - half), tuple(c + half), pick())

    def rect(place):
        plane_axis = int(rng.integers(3))
        a0, a1 = [a for a in range(3) if a != plane_axis]
        c, half = place(point()), size * u(0.3, 3.0, 2)
        return sc.rect(a0, a1, c[a0] - half[0], c[a0] + half[0], c[a1] - half[1], c[a1] + half[1], float(c[plane_axis]), pick())

    for _ in range(int(rng.integers(6, 14))):
        kind = int(rng.integers(8))
        t, place = transform()
        if kind <= 1:
            sc.add_prim(sphere(place), t)
        elif kind == 2:
            sc.add_prim(box(place), t)
        elif kind == 3:
            sc.add_prim(rect(place), t)
        elif kind <= 5:
            n = int(rng.integers(1, 70))
            prims = [sphere(place, rmax=1.2) if rng.random() < 0.7 else box(place, smax=1.5) for _ in range(n)]
            order = rng.permutation(n)  # arbitrary topology: the tree is a median split over a random order
            sc.add_bvh([prims[i] for i in order], t)
        elif kind == 6:
            sc.add_medium(sphere(place, rmax=3.5) if rng.random() < 0.5 else box(place, smax=5.0), float(u(0.05, 1.5)) / size, isos[int(rng.integers(2))], t)
        else:
            c = point()
            r = size * float(u(1.0, 2.5))
            glass = sc.dielectric(1.5)
            sc.add_prim(sc.sphere(tuple(c), r, glass))
            sc.add_prim(sc.sphere(tuple(c), -float(u(0.3, 0.9)) * r, glass))  # hollow glass: negative radius (worlds.rs:54)
    if rng.random() < 0.5:
        sc.add_medium(sc.sphere(tuple(shift + size * np.array([0.0, 0.0, -15.0])), 80.0 * size, mats[0]), float(u(0.002, 0.02)) / size, isos[0])  # fog around everything
    return sc, lo, hi


def random_rays(seed, lo, hi, n):
    rng = np.random.default_rng(seed + 1000)
    rays = np.zeros((n, 6))
    ext = hi - lo
    target = rng.uniform(lo - 0.04 * ext, hi + 0.04 * ext, size=(n, 3))
    outside = rng.uniform(lo - 1.2 * ext, hi + 1.2 * ext, size=(n, 3))
    inside = rng.uniform(lo, hi, size=(n, 3))  # origins among (and inside) the objects
    origin = np.where((np.arange(n) % 3 == 0)[:, None], inside, outside)
    d = target - origin
    d *= rng.uniform(0.05, 30.0, size=(n, 1)) / np.linalg.norm(d, axis=1)[:, None]  # not normalised, as the reference's rays (any length)
    rays[:, :3], rays[:, 3:] = origin, d
    psd = np.zeros((n, 3), dtype=np.uint32)
    psd[:, 0] = rng.integers(0, 2 ** 31, size=n)
    psd[:, 1] = rng.integers(0, 10000, size=n)
    psd[:, 2] = 4 * rng.integers(1, 40, size=n) + (rng.integers(0, 4, size=n) * (rng.random(n) < 0.1))
    return rays, psd


# the same scenes at the scale of the reference worlds, a thousand times smaller (t_min = 0.001 in play), far from the origin
# (f32 culling entries 0.06 .. 0.25 coarse against spheres of radius 0.2: the outward rounding must hold) and large and far
PLACES = {"unit": (1.0, (0.0, 0.0, 0.0)), "tiny": (1e-3, (0.0, 0.0, 0.0)), "far": (1.0, (1e6, -2e6, 5e5)), "large": (300.0, (1e5, 1e5, 1e5))}


# (the fast build's f32 geometry is not meant for coordinates of 1e6 or sizes of 1e-3: it runs the "unit" scenes only)
@pytest.mark.parametrize("build,place", [(b, p) for b in BUILDS for p in sorted(PLACES) if b != "fp32" or p == "unit"])
def test_random_scenes_one_bounce_against_the_oracle(dbg, oracle, build, place):
    m = dbg
    f64 = build != "fp32"
    n_hit = n_scatter = n_total = n_soft = 0
    for seed in range(12):
        sc, lo, hi = random_scene(m, seed, *PLACES[place])
        scene = sc.to_ctypes()
        rays, psd = random_rays(seed, lo, hi, N_RAYS)
        orc = OracleScene(oracle, scene)
        e = orc.probe_rays(dc.SEED + seed, rays, psd)
        orc.close()
        ctx = context(m, scene, build)
        g = np.zeros((N_RAYS, 24))
        check(m, m.lib().mlrt_debug_probe_rays(ctx, dc.SEED + seed, N_RAYS, ptr(rays), ptr(psd), ptr(g)))
        m.lib().mlrt_context_destroy(ctx)
        flags = [0, 10, 11, 21, 22]  # hit, front_face, scattered, draw index, material
        same = np.all(g[:, flags] == e[:, flags], axis=1)
        hit = (e[:, 0] == 1) & same
        scale = np.maximum(1.0, np.abs(e[:, 1:5]).max(axis=1))
        rel = 1e-12 if f64 else 3e-5
        close = np.ones(N_RAYS, dtype=bool)
        close &= ~hit | (np.abs(g[:, 1] - e[:, 1]) <= (1e-12 if f64 else 2e-6) * np.maximum(1.0, np.abs(e[:, 1])))
        close &= ~hit | (np.abs(g[:, 2:8] - e[:, 2:8]).max(axis=1) <= rel * scale)
        uv_known = hit & ((g[:, 8] != 0.0) | (g[:, 9] != 0.0))  # (the device computes u, v only where an image texture reads them)
        close &= ~uv_known | (np.abs(g[:, 8:10] - e[:, 8:10]).max(axis=1) <= (1e-12 if f64 else 3e-5))
        scattered = hit & (e[:, 11] == 1)
        close &= ~scattered | (np.abs(g[:, 12:18] - e[:, 12:18]).max(axis=1) <= rel * np.maximum(scale, np.abs(e[:, 15:18]).max(axis=1)))
        close &= ~hit | (np.abs(g[:, 18:21] - e[:, 18:21]).max(axis=1) <= (1e-12 if f64 else 3e-5) * np.maximum(1.0, np.abs(e[:, 18:21]).max(axis=1)))
        bad = ~(same & close)
        if f64:
            first = np.flatnonzero(bad)[:3]
            assert not bad.any(), (build, seed, int(bad.sum()), [(int(i), rays[i].tolist(), g[i].tolist(), e[i].tolist()) for i in first])
        n_soft += int(bad.sum())
        n_total += N_RAYS; n_hit += int((e[:, 0] == 1).sum()); n_scatter += int((e[:, 11] == 1).sum())
    assert n_hit > 0.4 * n_total and n_scatter > 0.3 * n_total, (n_hit, n_scatter, n_total)  # the rays do exercise the scenes
    assert n_soft <= 0.003 * n_total, (build, n_soft, n_total)


# ======================================================================================================================
# Whole frames of random scenes: the megakernel itself (lane state machine, batches, tile queue, candidate lists, media
# sequences, depth limit, accumulation, to_rgb) through the C ABI of the RELEASE library against the oracle's render of the
# same flattened scene with the same Philox streams.  Criteria as tests/test_gpu_parity.py: f64 8-bit image equal - exactly for the
# sphere-only scenes, whose device path calls no libm function; <= 0.05 % of the channels where sin / log / acos / atan2 are in play
# (small frames: one flipped sample is a visible share) -, fast build by RMSE.
def cam19(c):
    return list(c.origin) + list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u) + list(c.v) + [c.lens_radius]


def render_scene_both(mlrt, oracle, scene, cam, width, height, spp, max_depth, seed, background, precision, opts=None):
    import ctypes as C
    orc = OracleScene(oracle, scene)
    ref, _ = orc.render(cam19(cam), width, height, spp, max_depth=max_depth, seed=seed, background=background)
    orc.close()
    p = mlrt.Params()
    p.image_width, p.image_height, p.samples_per_pixel, p.max_depth, p.seed, p.background_kind = width, height, spp, max_depth, seed, background
    p.background_top[:] = (0.5, 0.7, 1.0); p.background_bottom[:] = (1.0, 1.0, 1.0)
    ctx = C.c_void_p()
    o = opts if opts is not None else mlrt.options()
    rc = mlrt.lib().mlrt_context_create_opts(C.byref(scene), 0, precision, C.byref(o), C.byref(ctx))
    assert rc == 0, mlrt.lib().mlrt_last_error()
    got = np.zeros((height, width, 3), dtype=np.int32)
    rc = mlrt.lib().mlrt_context_render(ctx, C.byref(cam), C.byref(p), None, height, got.ctypes.data_as(C.c_void_p), None, None)
    assert rc == 0, mlrt.lib().mlrt_last_error()
    info = (C.c_int64 * 4)()
    mlrt.lib().mlrt_context_info(ctx, info)
    mlrt.lib().mlrt_context_destroy(ctx)
    return got, ref, int(info[1])  # info[1]: scene resident in LDS


def diff(got, ref):
    d = (got - ref).astype(np.float64)
    return float(np.mean(d != 0)), float(np.sqrt(np.mean(d * d)) / 255.0)


def random_camera(mlrt, rng, lo, hi, aspect, above_ground=False):
    centre = 0.5 * (lo + hi)
    lookat = centre + rng.uniform(-0.3, 0.3, 3) * (hi - lo)
    away = rng.normal(size=3)
    away /= np.linalg.norm(away)
    lookfrom = lookat + away * rng.uniform(0.2, 2.5) * float(np.linalg.norm(hi - lo))  # sometimes among the objects, sometimes far outside
    if above_ground:
        lookfrom[1] = lo[1] + abs(lookfrom[1] - lo[1]) + 0.05
    lens = rng.random() < 0.4
    focus = float(np.linalg.norm(lookfrom - lookat))
    return mlrt.camera_new(tuple(lookfrom), tuple(lookat), (0.0, 1.0, 0.0), float(rng.uniform(15.0, 80.0)), aspect, float(rng.uniform(0.02, 0.5)) if lens else 0.0, focus)


@pytest.mark.parametrize("precision_name", ["fp64", "fp32"])
def test_random_full_scenes_whole_frames(mlrt, oracle, precision_name):
    """FULL kernel variant: the random scenes of the one-bounce test above, random cameras (pinhole and lens), both backgrounds."""
    precision = mlrt.PRECISION_FP64 if precision_name == "fp64" else mlrt.PRECISION_FP32
    worst = (0.0, 0.0, -1)
    for seed in range(8):
        sc, lo, hi = random_scene(mlrt, seed)
        rng = np.random.default_rng(500 + seed)
        w, h, spp = 64 + 8 * seed, 48, [8, 33, 64, 130][seed % 4]
        cam = random_camera(mlrt, rng, lo, hi, w / h)
        got, ref, _ = render_scene_both(mlrt, oracle, sc.to_ctypes(), cam, w, h, spp, [50, 5, 1, 12][seed % 4], 1000 + seed, int(seed % 4 != 0), precision)
        frac, rmse = diff(got, ref)
        assert ref.any()
        worst = max(worst, (frac, rmse, seed))
        if precision_name == "fp64":
            assert frac <= 5e-4, (seed, frac, rmse)
        else:
            assert rmse <= 1.0e-2 / np.sqrt(spp) + 2e-4, (seed, frac, rmse)


def sphere_scene(mlrt, seed, shift=(0.0, 0.0, 0.0)):
    """Solid-colour spheres only: packs as the LEAN variant (sphere-only step, primary candidate lists)."""
    rng = np.random.default_rng(9000 + seed)
    shift = np.asarray(shift, dtype=np.float64)
    sc = dc.MiniScene(mlrt)
    u = rng.uniform
    mats = [sc.lambertian(sc.solid(*u(0.1, 0.9, 3))) for _ in range(6)] + [sc.metal(tuple(u(0.4, 1.0, 3)), float(u(0.0, 0.6))) for _ in range(3)] + [sc.dielectric(1.5)]
    n = [1, 2, 7, 40, 300, 1500][seed % 6]
    extent = 2.0 + n ** (1.0 / 3.0) * 1.5
    lo, hi = np.array([-extent, 0.0, -extent]) + shift, np.array([extent, 0.6 * extent, extent]) + shift
    prims = [sc.sphere(tuple(shift + (0.0, -1000.0, 0.0)), 1000.0, mats[0])]  # the ground, as `random` has it
    for _ in range(n):
        prims.append(sc.sphere(tuple(u(lo, hi)), float(u(0.1, 0.9)), mats[int(rng.integers(len(mats)))]))
    if seed % 3 == 0:
        sc.add_bvh([prims[i] for i in rng.permutation(len(prims))])
    else:  # some at the top level, the rest in one or two BVHs
        k = int(rng.integers(1, min(6, len(prims)) + 1))
        for p in prims[:k]:
            sc.add_prim(p)
        rest = prims[k:]
        if rest:
            half = len(rest) // 2 if seed % 3 == 2 else 0
            if half:
                sc.add_bvh(rest[:half])
            sc.add_bvh(rest[half:])
    return sc, lo, hi


@pytest.mark.parametrize("precision_name", ["fp64", "fp32"])
def test_random_sphere_scenes_whole_frames(mlrt, oracle, precision_name):
    """LEAN kernel variant with the primary candidate lists: 1 .. 1500 random spheres, cameras outside and among them (a frustum
    apex inside a sphere's box, lens or pinhole), frame sizes around the tile planner's thresholds; the same frames with the lists
    switched off and with the scene read from HBM must be bit-identical."""
    precision = mlrt.PRECISION_FP64 if precision_name == "fp64" else mlrt.PRECISION_FP32
    for seed in range(12):
        # (f64, the last four scenes: 1e5 .. 1e6 from the origin - candidate lists and culling entries at coarse f32 resolution)
        far = precision_name == "fp64" and seed >= 8
        sc, lo, hi = sphere_scene(mlrt, seed, (3e5, 1e5, -1e6) if far else (0.0, 0.0, 0.0))
        rng = np.random.default_rng(700 + seed)
        w, h = [(96, 64), (33, 17), (128, 40), (60, 60)][seed % 4]
        spp = [16, 127, 64, 200][seed % 4]
        cam = random_camera(mlrt, rng, lo, hi, w / h, above_ground=seed != 7)  # (seed 7: the camera inside the ground sphere, a black frame)
        scene = sc.to_ctypes()
        got, ref, lds = render_scene_both(mlrt, oracle, scene, cam, w, h, spp, 50 if seed % 5 else 3, 2000 + seed, 1, precision)
        frac, rmse = diff(got, ref)
        assert lds == 1 or len(sc.spheres) > 1000  # (1500 spheres with f64 payload slots: read from HBM / L2)
        if precision_name == "fp64":
            assert frac == 0.0, (seed, frac, rmse)  # sphere-only scenes: no libm call on the device path, the frame is the oracle's
        else:
            assert rmse <= 1e-3, (seed, frac, rmse)
        if seed % 2 == 0:
            for kw in (dict(primary_lists=0), dict(lds_scene=0), dict(sah=0, leaf_wrap=0)):
                other, _, _ = render_scene_both(mlrt, oracle, scene, cam, w, h, spp, 50 if seed % 5 else 3, 2000 + seed, 1, precision, opts=mlrt.options(**kw))
                assert np.array_equal(other, got), (seed, kw)
