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


def random_scene(m, seed):
    """(scene, lo, hi): objects inside the box [lo, hi]."""
    rng = np.random.default_rng(seed)
    sc = dc.base(m)
    u = rng.uniform
    mats = [sc.lambertian(sc.t_white), sc.lambertian(sc.t_image), sc.lambertian(sc.t_checker), sc.lambertian(sc.t_noise), sc.light(sc.t_light)]
    mats += [sc.metal(tuple(u(0.3, 1.0, 3)), float(u(0.0, 1.0))) for _ in range(3)]
    mats += [sc.metal((0.9, 0.8, 0.7), 0.0), sc.dielectric(1.5), sc.dielectric(float(u(1.05, 2.4)))]
    isos = [sc.isotropic(sc.solid(*u(0.1, 1.0, 3))), sc.isotropic(sc.t_checker)]
    lo, hi = np.array([-12.0, -8.0, -30.0]), np.array([12.0, 8.0, -6.0])
    pick = lambda: mats[int(rng.integers(len(mats)))]  # noqa: E731
    point = lambda: u(lo, hi)  # noqa: E731

    def sphere(c=None, rmax=2.5):
        return sc.sphere(tuple(point() if c is None else c), float(u(0.2, rmax)), pick())

    def box(c=None, smax=3.0):
        c = point() if c is None else np.asarray(c)
        half = u(0.15, smax / 2, 3)
        return sc.box(tuple(c - half), tuple(c + half), pick())

    def rect():
        plane_axis = int(rng.integers(3))
        a0, a1 = [a for a in range(3) if a != plane_axis]
        c, half = point(), u(0.3, 3.0, 2)
        return sc.rect(a0, a1, c[a0] - half[0], c[a0] + half[0], c[a1] - half[1], c[a1] + half[1], float(c[plane_axis]), pick())

    def transform():
        k = int(rng.integers(4))
        if k == 0:
            return -1
        axis = int(rng.integers(3)) if k >= 2 else None
        off = tuple(u(-3.0, 3.0, 3)) if k != 2 else None
        return sc.transform(rotate_axis=axis, degrees=float(u(-80.0, 80.0)), offset=off)

    for _ in range(int(rng.integers(6, 14))):
        kind = int(rng.integers(8))
        if kind <= 1:
            sc.add_prim(sphere(), transform())
        elif kind == 2:
            sc.add_prim(box(), transform())
        elif kind == 3:
            sc.add_prim(rect(), transform())
        elif kind <= 5:
            n = int(rng.integers(1, 70))
            prims = [sphere(rmax=1.2) if rng.random() < 0.7 else box(smax=1.5) for _ in range(n)]
            order = rng.permutation(n)  # arbitrary topology: the tree is a median split over a random order
            sc.add_bvh([prims[i] for i in order], transform())
        elif kind == 6:
            sc.add_medium(sphere(rmax=3.5) if rng.random() < 0.5 else box(smax=5.0), float(u(0.05, 1.5)), isos[int(rng.integers(2))], transform())
        else:
            c = point()
            r = float(u(1.0, 2.5))
            glass = sc.dielectric(1.5)
            sc.add_prim(sc.sphere(tuple(c), r, glass))
            sc.add_prim(sc.sphere(tuple(c), -float(u(0.3, 0.9)) * r, glass))  # hollow glass: negative radius (worlds.rs:54)
    if rng.random() < 0.5:
        sc.add_medium(sc.sphere((0.0, 0.0, -15.0), 80.0, mats[0]), float(u(0.002, 0.02)), isos[0])  # fog around everything
    return sc, lo, hi


def random_rays(seed, lo, hi, n):
    rng = np.random.default_rng(seed + 1000)
    rays = np.zeros((n, 6))
    target = rng.uniform(lo - 1.0, hi + 1.0, size=(n, 3))
    outside = rng.uniform([-40.0, -25.0, -60.0], [40.0, 25.0, 25.0], size=(n, 3))
    inside = rng.uniform(lo, hi, size=(n, 3))  # origins among (and inside) the objects
    origin = np.where((np.arange(n) % 3 == 0)[:, None], inside, outside)
    d = target - origin
    d *= rng.uniform(0.05, 30.0, size=(n, 1)) / np.linalg.norm(d, axis=1)[:, None]  # not normalised, as the reference's rays
    rays[:, :3], rays[:, 3:] = origin, d
    psd = np.zeros((n, 3), dtype=np.uint32)
    psd[:, 0] = rng.integers(0, 2 ** 31, size=n)
    psd[:, 1] = rng.integers(0, 10000, size=n)
    psd[:, 2] = 4 * rng.integers(1, 40, size=n) + (rng.integers(0, 4, size=n) * (rng.random(n) < 0.1))
    return rays, psd


@pytest.mark.parametrize("build", BUILDS)
def test_random_scenes_one_bounce_against_the_oracle(dbg, oracle, build):
    m = dbg
    f64 = build != "fp32"
    n_hit = n_scatter = n_total = n_soft = 0
    for seed in range(12):
        sc, lo, hi = random_scene(m, seed)
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
    assert n_hit > 0.5 * n_total and n_scatter > 0.3 * n_total, (n_hit, n_scatter, n_total)  # the rays do exercise the scenes
    assert n_soft <= 0.003 * n_total, (build, n_soft, n_total)
