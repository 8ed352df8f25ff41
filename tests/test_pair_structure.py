"""The pair structure the packer builds for sphere-only scenes (mu-lambda-raytracer_amd/csrc/mlrt_device.hpp DPair, built by
pack_pairs in mlrt_api.cpp; walked by the PAIR variant of the render kernel), checked on the CPU through the test hook
mlrt_debug_pack_pairs - no device call:

* structure: every sphere is a leaf exactly once, numbered in the reference's visiting order (hittable.rs:57-66 list order,
  bhv.rs:152-163 left before right); every child box encloses the spheres below it (rounded OUTWARD: the culling test may
  never lose a sphere); `stack_levels` is the longest chain of inner nodes, and never above the builder's cap;
* the walk itself, restated in numpy as test infrastructure (near child first, far child pushed, a hit at an equal distance
  only from a later primitive): its closest hit equals the brute-force closest hit over all spheres in the reference's order
  - `t <= closest_so_far` accepted, so the LAST of equal distances wins (shapes.rs:66-73 with hittable.rs:58-66) - for every
  ray, including scenes of coincident spheres where ties are the rule;
* the stack never holds more than `stack_levels` references.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

import directed_cases as dc

mlrt = importlib.import_module("mu-lambda-raytracer_amd")

PAIR_DT = np.dtype([("a", "<f4", 6), ("b", "<f4", 6), ("ref_a", "<i4"), ("ref_b", "<i4"), ("pad", "<i4", 2)])
LEAF, END, LEVEL_CAP = -32768, -1, 24


def pack(scene, precision):
    L = mlrt.lib()
    L.mlrt_debug_pack_pairs.argtypes = [C.POINTER(mlrt.Scene), C.c_int32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]
    blob, info = C.POINTER(C.c_uint8)(), (C.c_int64 * 8)()
    rc = L.mlrt_debug_pack_pairs(scene, precision, C.byref(blob), info)
    if rc != 0:
        return rc, None
    nbytes, off_pairs, n_pairs, off_prims, off_mat, n_prims, levels, _ = [int(x) for x in info]
    raw = np.ctypeslib.as_array(blob, shape=(nbytes,)).copy()
    L.mlrt_free(blob)
    real = np.float32 if precision == mlrt.PRECISION_FP32 else np.float64
    pairs = np.frombuffer(raw.tobytes(), dtype=PAIR_DT, count=n_pairs, offset=off_pairs)
    prims = np.frombuffer(raw.tobytes(), dtype=real, count=4 * n_prims, offset=off_prims).reshape(n_prims, 4).astype(np.float64)
    mats = np.frombuffer(raw.tobytes(), dtype=np.uint16, count=n_prims, offset=off_mat)
    return 0, dict(pairs=pairs, prims=prims, mats=mats, levels=levels)


def reference_order(sc):
    """Sphere indices in the order the reference visits them: objects in list order, a BVH's leaves left to right."""
    order = []
    for o in sc.objects:
        if o["kind"] == 0:
            order.append(o["prim"][1])
            continue
        root = sc.bvhs[o["bvh"]]["root"]

        def leaves(child):
            if child < 0:
                order.append(sc.prim_refs[~child][1])
                return
            leaves(sc.bvh_nodes[child]["left"]); leaves(sc.bvh_nodes[child]["right"])
        if root != -(2 ** 31):
            leaves(root)
    return order


def check_structure(P, spheres_in_order, exact_f64):
    pairs, prims, levels = P["pairs"], P["prims"], P["levels"]
    want = np.array([[*s["center"], s["radius"]] for s in spheres_in_order], dtype=np.float64)
    if exact_f64:
        assert np.array_equal(prims, want)
    else:
        assert np.allclose(prims, want, rtol=1e-6, atol=0)
    seen = np.zeros(len(prims), dtype=int)
    deepest = [0]

    def visit(ref, level):
        """Returns the bounds of the subtree; `level` = inner nodes above it."""
        if ref < 0:
            prim = ref - LEAF
            assert 0 <= prim < len(prims)
            seen[prim] += 1
            c, r = want[prim, :3], abs(want[prim, 3])  # (the scene's f64 sphere: the fast build's f32 copy may sit an ulp outside)
            return c - r, c + r
        deepest[0] = max(deepest[0], level + 1)
        node = pairs[ref]
        lo_a, hi_a = visit(int(node["ref_a"]), level + 1)
        lo_b, hi_b = visit(int(node["ref_b"]), level + 1)
        for (lo, hi), box in (((lo_a, hi_a), node["a"]), ((lo_b, hi_b), node["b"])):
            assert np.all(box[:3].astype(np.float64) <= lo) and np.all(box[3:].astype(np.float64) >= hi), "a child box must enclose its spheres"
        return np.minimum(lo_a, lo_b), np.maximum(hi_a, hi_b)

    visit(0, 0)
    if len(prims) > 1:
        assert np.all(seen == 1), "every sphere is a leaf exactly once"
    else:
        assert seen[0] == 2  # a single sphere: both children of the one node
    assert deepest[0] == levels <= LEVEL_CAP


def slab(box, o, inv, closest):
    t0, t1 = (box[:3] - o) * inv, (box[3:] - o) * inv
    tn, tf = max(np.max(np.minimum(t0, t1)), 0.001), np.min(np.maximum(t0, t1))
    return (not (tf * 1.000001 <= tn)) and not (closest * 1.000001 <= tn), tn


def sphere_first_root(o, d, c, r, tmin, tmax):
    oc = o - c
    a, hb, cc = d @ d, oc @ d, oc @ oc - r * r
    disc = hb * hb - a * cc
    if disc < 0:
        return None
    sq = np.sqrt(disc)
    root = (-hb - sq) / a
    if root < tmin or tmax < root:
        root = (-hb + sq) / a
        if root < tmin or tmax < root:
            return None
    return root


def walk(P, o, d):
    """The device's walk (mlrt_kernel.hip, PAIR step + sphere batch) for one ray: (hit prim or -1, t, node visits, deepest stack)."""
    pairs, prims = P["pairs"], P["prims"]
    with np.errstate(divide="ignore"):
        inv = 1.0 / np.where(np.abs(d) < 1e-30, np.copysign(1e-30, d), d)
    closest, hit, stack, node, visits, deepest = np.inf, -1, [END], 0, 0, 0
    while True:
        if node >= 0:
            visits += 1
            n = pairs[node]
            (ha, tna), (hb, tnb) = slab(n["a"].astype(np.float64), o, inv, closest), slab(n["b"].astype(np.float64), o, inv, closest)
            b_first = hb and (not ha or tnb < tna)
            first, second = (int(n["ref_b"]), int(n["ref_a"])) if b_first else (int(n["ref_a"]), int(n["ref_b"]))
            if ha and hb:
                stack.append(second)
                deepest = max(deepest, len(stack) - 1)
            elif not ha and not hb:
                first = stack.pop()
            node = first
        elif node == END:
            return hit, closest, visits, deepest
        else:
            prim = node - LEAF
            t = sphere_first_root(o, d, prims[prim, :3], prims[prim, 3], 0.001, closest)
            if t is not None and (t < closest or prim > hit):
                closest, hit = t, prim
            node = stack.pop()


def brute(P, o, d):
    """HittableList::hit over the spheres in the reference's order (hittable.rs:53-69)."""
    closest, hit = np.inf, -1
    for k, s in enumerate(P["prims"]):
        t = sphere_first_root(o, d, s[:3], s[3], 0.001, closest)
        if t is not None:
            closest, hit = t, k
    return hit, closest


def random_rays(rng, prims, n):
    lo, hi = (prims[:, :3] - np.abs(prims[:, 3:4])).min(0), (prims[:, :3] + np.abs(prims[:, 3:4])).max(0)
    small = np.abs(prims[:, 3]) < 50.0  # (aim at the small spheres: a ground sphere of radius 1000 swamps the box)
    if small.any():
        lo, hi = (prims[small, :3] - np.abs(prims[small, 3:4])).min(0), (prims[small, :3] + np.abs(prims[small, 3:4])).max(0)
    ext = np.maximum(hi - lo, 1e-3)
    target = rng.uniform(lo, hi, size=(n, 3))
    origin = np.where((np.arange(n) % 3 == 0)[:, None], rng.uniform(lo, hi, size=(n, 3)), rng.uniform(lo - ext, hi + ext, size=(n, 3)))
    d = target - origin
    d *= rng.uniform(0.05, 30.0, size=(n, 1)) / np.linalg.norm(d, axis=1)[:, None]
    d[::17, 0] = 0.0  # axis-parallel components
    return origin, d


def check_walk(P, n_rays, seed):
    rng = np.random.default_rng(seed)
    origin, d = random_rays(rng, P["prims"], n_rays)
    visits = 0
    for o, dd in zip(origin, d):
        hit, t, v, deepest = walk(P, o, dd)
        want_hit, want_t = brute(P, o, dd)
        assert (hit, t) == (want_hit, want_t)
        assert deepest <= P["levels"]
        visits += v
    return visits / n_rays


@pytest.mark.parametrize("precision", [mlrt.PRECISION_FP32, mlrt.PRECISION_FP64])
def test_random_world(precision):
    """BASELINE configs[0..2]'s world: 486 spheres in one BHV."""
    w = mlrt.World("random", 42)
    rc, P = pack(w.scene, precision)
    assert rc == 0
    s = w.scene.contents
    assert len(P["prims"]) == s.n_spheres == 486
    sc_like = type("S", (), {})()
    sc_like.objects = [dict(kind=s.objects[i].kind, bvh=s.objects[i].bvh, prim=(s.objects[i].prim.kind, s.objects[i].prim.index)) for i in range(s.n_objects)]
    sc_like.bvhs = [dict(root=s.bvhs[i].root) for i in range(s.n_bvhs)]
    sc_like.bvh_nodes = [dict(left=s.bvh_nodes[i].left, right=s.bvh_nodes[i].right) for i in range(s.n_bvh_nodes)]
    sc_like.prim_refs = [(s.prim_refs[i].kind, s.prim_refs[i].index) for i in range(s.n_prim_refs)]
    order = reference_order(sc_like)
    spheres = [dict(center=tuple(s.spheres[i].center), radius=s.spheres[i].radius) for i in order]
    check_structure(P, spheres, precision == mlrt.PRECISION_FP64)
    assert np.array_equal(P["mats"], [s.spheres[i].material for i in order])
    mean_visits = check_walk(P, 400, 5)
    assert mean_visits < 40  # (a sanity bound: ~10 for this scene)
    w.close()


def sphere_scene(rng, n, layout):
    sc = dc.MiniScene(mlrt)
    mats = [sc.lambertian(sc.solid(0.5, 0.5, 0.5)), sc.metal((0.8, 0.8, 0.8), 0.1), sc.dielectric(1.5)]
    prims = []
    for k in range(n):
        if layout == "cloud":
            c, r = rng.uniform(-20, 20, 3), float(rng.uniform(0.1, 2.0))
        elif layout == "coincident":  # a handful of distinct spheres, each many times: ties everywhere
            j = k % 5
            c, r = np.array([3.0 * j, 0.0, -10.0]), 1.0 + 0.25 * (j % 2)
        elif layout == "chain":  # nested shells along a line: a surface-area split peels one sphere per level
            c, r = np.array([1e-3 * k, 0.0, 0.0]), 1.5 ** (k % 60) * (1.0 + k // 60)
        else:  # "hollow": negative radii (worlds.rs:54)
            c, r = rng.uniform(-10, 10, 3), float(rng.uniform(0.3, 2.0)) * (-1.0 if k % 2 else 1.0)
        prims.append(sc.sphere(tuple(c), r, mats[k % 3]))
    # part of them as top-level primitives, the rest in BVHs of arbitrary order
    cut = int(rng.integers(0, n + 1)) if n > 2 else n
    for p in prims[:cut]:
        sc.add_prim(p)
    rest = prims[cut:]
    if rest:
        half = len(rest) // 2
        sc.add_bvh([rest[i] for i in rng.permutation(half)] if half else [])
        sc.add_bvh(rest[half:])
    return sc


@pytest.mark.parametrize("layout,n", [("cloud", 1), ("cloud", 2), ("cloud", 3), ("cloud", 700), ("coincident", 60), ("chain", 200), ("hollow", 40)])
def test_synthetic_scenes(layout, n):
    rng = np.random.default_rng(1000 + n)
    sc = sphere_scene(rng, n, layout)
    spheres = [sc.spheres[i] for i in reference_order(sc)]
    for precision in (mlrt.PRECISION_FP32, mlrt.PRECISION_FP64):
        rc, P = pack(sc.to_ctypes(), precision)
        assert rc == 0
        if layout == "chain" and precision == mlrt.PRECISION_FP32:
            continue  # (f32 cannot hold radii of 1.5^59 next to offsets of 1e-3; the f64 pack is the check)
        check_structure(P, spheres, precision == mlrt.PRECISION_FP64)
        check_walk(P, 150 if n > 100 else 300, n)


def test_outside_the_limits():
    """Worlds with anything but spheres and solid colours, and more spheres than a 16-bit reference addresses, keep the threaded list."""
    w = mlrt.World("cornell_box", 42)
    rc, _ = pack(w.scene, mlrt.PRECISION_FP64)
    assert rc == mlrt.MLRT_ERR_UNSUPPORTED
    w.close()
    sc = dc.MiniScene(mlrt)
    mat = sc.lambertian(sc.solid(0.5, 0.5, 0.5))
    rng = np.random.default_rng(3)
    sc.add_bvh([sc.sphere(tuple(rng.uniform(-100, 100, 3)), 0.5, mat) for _ in range(32768)])
    rc, _ = pack(sc.to_ctypes(), mlrt.PRECISION_FP64)
    assert rc == mlrt.MLRT_ERR_UNSUPPORTED
