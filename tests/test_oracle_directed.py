"""CPU: the committed directed vectors (tests/golden/directed_vectors.npz, made by tests/golden/make_directed_vectors.py) are what
the oracle computes today, and the reference's OWN known answers are among them with the values the reference's unit tests
assert (bhv.rs:168-221, shapes.rs:200-214)."""
import os

import numpy as np
import pytest

import directed_cases as dc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "directed_vectors.npz"))


@pytest.fixture(scope="module")
def fresh(mlrt, oracle):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_directed_vectors
    return make_directed_vectors.evaluate(mlrt, oracle)


def test_oracle_reproduces_the_committed_vectors(fresh):
    assert sorted(fresh) == sorted(GOLD.files)
    for key in GOLD.files:
        a, b = GOLD[key], fresh[key]
        if a.dtype.kind in "US":
            assert list(a) == list(b), key
        else:
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), key


def test_reference_aabb_unit_tests_are_in_the_vectors():
    """bhv.rs:168-221, t in [0, inf): minmax hit, edge_parallel hit, edge_parallel_outside miss, face_parallel hit,
    face_parallel_outside miss - for the box given as (min, max) and as (max, min)."""
    labels = [str(x) for x in GOLD["aabb_labels"][:10]]
    assert all(l.startswith("bhv.rs test_") for l in labels)
    assert list(GOLD["expect_aabb_tmin0"][:10]) == [1, 1, 0, 1, 0] * 2
    # the device's t_min is 0.001 (raytrace.rs:90): same answers
    assert list(GOLD["expect_aabb"][:10]) == [1, 1, 0, 1, 0] * 2


def test_reference_sphere_uv_unit_test_is_in_the_vectors():
    """shapes.rs:200-214 (exact equality in the reference): the six axis points of the unit sphere, reached by rays."""
    e = GOLD["expect_primitives"][:6]
    assert all(str(l).startswith("sphere_uv") for l in GOLD["labels_primitives"][:6])
    assert [tuple(r[8:10]) for r in e] == [(0.5, 0.5), (0.5, 1.0), (0.25, 0.5), (0.0, 0.5), (0.5, 0.0), (0.75, 0.5)]
    assert np.all(e[:, 0] == 1) and np.all(e[:, 1] == 1.5)


def test_to_rgb_edge_values():
    """raytrace.rs:59-68: gamma 2, clamp to [0, 0.99999999], 255.999 * c truncated; `NaN as i32` is 0 and negatives clamp to 0."""
    vals, out = GOLD["to_rgb_vals"], GOLD["expect_to_rgb_spp1"]
    assert out[list(vals).index(0.25)] == 127 and out[list(vals).index(1.0)] == 255 and out[list(vals).index(0.0)] == 0
    assert out[-1] == 0 and np.isnan(vals[-1])  # NaN
    assert all(out[i] == 0 for i, v in enumerate(vals) if v < 0)
    assert all(out[i] == 255 for i, v in enumerate(vals) if v >= 1.0)


def test_a_block_with_reversed_corners_is_rejected(mlrt):
    """Block::new(p0, p1) with p0 > p1 builds degenerate rects in the reference (aarects.rs:39, quirk Q2): the C ABI refuses it."""
    import ctypes as C
    sc = dc.base(mlrt)
    sc.add_prim(sc.box((1.0, 1.0, 1.0), (0.0, 2.0, 2.0), sc.lambertian(sc.t_white)))
    ctx = C.c_void_p()
    rc = mlrt.lib().mlrt_context_create(C.byref(sc.to_ctypes()), 0, mlrt.PRECISION_FP64, C.byref(ctx))
    assert rc == mlrt.MLRT_ERR_INVALID and b"box min" in mlrt.lib().mlrt_last_error()


@pytest.mark.parametrize("name,width,height,spp", [("simple", 40, 24, 4), ("random", 48, 32, 4), ("random_chk", 32, 24, 4), ("simple_light", 32, 24, 8),
                                                   ("cornell_box", 24, 24, 8), ("cornell_smoke", 24, 24, 8), ("earth", 32, 24, 4), ("debug_perlin", 24, 24, 4),
                                                   ("final_scene", 24, 24, 8)])
def test_scene_oracle_is_the_world_oracle(mlrt, oracle, earthmap, name, width, height, spp):
    """The oracle over an arbitrary FLATTENED scene (oracle_scene.cpp: the checker of tests/test_gpu_fuzz.py and of the directed
    probes) renders the host's flattened form of every kind of world to the very same colour sums as the oracle's own world
    builders (oracle.cpp build_world: the implementation the golden images and the reference's known answers pin) - both RNG
    contracts.  So a scene handed over the C ABI means to the checker what the reference's world means."""
    from oracle_binding import RNG_PCG_ROW, RNG_PHILOX_SAMPLE
    earth = earthmap if name in ("earth", "final_scene") else None
    pw = mlrt.World(name, 42, earth)
    ow = oracle.world(name, 42, earth)
    cam = mlrt.default_camera(pw, width / height, 0.05, None)
    cam19 = list(cam.origin) + list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) + list(cam.u) + list(cam.v) + [cam.lens_radius]
    p = mlrt.make_params(pw, width, height, spp)
    osc = oracle.scene(pw.scene.contents if hasattr(pw.scene, "contents") else pw.scene)
    for mode in (RNG_PHILOX_SAMPLE, RNG_PCG_ROW):
        ref_rgb, ref_acc, _, _ = ow.render(cam19, width, height, spp, max_depth=12, seed=42, rng_mode=mode, want_accum=True)
        got_rgb, got_acc = osc.render(cam19, width, height, spp, max_depth=12, seed=42, rng_mode=mode, background=int(p.background_kind),
                                      top=tuple(p.background_top), bottom=tuple(p.background_bottom), want_accum=True)
        assert np.array_equal(got_acc, ref_acc) and np.array_equal(got_rgb, ref_rgb), (name, mode)
    osc.close()
    ow.close()
    pw.close()
