"""Pins the CPU oracle to everything the reference offers for this path:
third-party generator known answers, the reference's own unit tests, the seed-42 scene known answers."""
import math

import numpy as np


# ---- rand_pcg 0.3.0 / rand_core 0.6.2 / rand 0.8.3 (sources absent from /root/reference; published algorithms) ----
def test_pcg64_published_demo_vector(oracle):
    # pcg64 demo: state 42, stream 54 (pcg-random.org reference output for pcg_setseq_128_xsl_rr_64)
    assert oracle.pcg64_new_stream(42, 54, 6) == [0x86B1DA1D72062B68, 0x1304AA46C9853D39, 0xA3670E9E0DD50358, 0xF9090E529A7DAE00,
                                                  0xC85B9FD837996F2C, 0x606121F8E3919196]


def test_pcg64_seed_from_u64_known_answers(oracle):
    state, inc = oracle.pcg64_state(42)
    assert state == 0x885A4FDB1D009FEBF6D2CE35380EF4F6
    assert inc == 0xF220DFE4A16E448D54F7B41875C88C2B
    assert oracle.pcg64_u64(42, 4) == [0x39FCB970A3001809, 0x3D3618972C55D911, 0xC2C5FA789A8B6A2D, 0x87207FF1E29660EC]
    assert oracle.pcg64_u64(43, 1) == [0x8675F2C1CD8D59BE]
    assert oracle.pcg64_uniform(42, 0.0, 1.0, 4) == [0.2265125179283134, 0.23910669031859944, 0.7608334106867409, 0.5278396573581516]


def test_gen_range_usize_matches_widening_multiply_rule(oracle):
    # UniformInt::sample_single: zone = (range << lz) - 1; accept when low64(v*range) <= zone
    raw = oracle.pcg64_u64(7, 64)
    expect, it = [], iter(raw)
    for _ in range(24):
        while True:
            v = next(it)
            m = v * 3
            if (m & ((1 << 64) - 1)) <= (3 << 62) - 1:
                expect.append(m >> 64)
                break
    assert oracle.pcg64_usize(7, 0, 3, 24) == expect
    assert all(0 <= x < 3 for x in expect)


def test_philox4x32_10_random123_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10
    assert oracle.philox([0, 0], [0, 0, 0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 2, [0xFFFFFFFF] * 4) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0xA4093822, 0x299F31D0], [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_stream_word_order(oracle):
    seed, pixel, sample = 0x0123456789ABCDEF, 77, 5
    u = oracle.philox_uniform(seed, pixel, sample, 9)
    key = [seed & 0xFFFFFFFF, seed >> 32]
    words = oracle.philox(key, [0, sample, pixel, 0]) + oracle.philox(key, [1, sample, pixel, 0]) + oracle.philox(key, [2, sample, pixel, 0])
    assert u == [(w >> 8) / 16777216.0 for w in words[:9]]


# ---- the reference's own unit tests for this path ----
def test_reference_aabb_tests(oracle):
    """bhv.rs:168-221 aabb_tests (box and its min/max-swapped twin, t in [0, inf))."""
    inf = math.inf
    boxes = [((1, 1, 1), (2, 2, 2)), ((2, 2, 2), (1, 1, 1))]
    cases = [((0, 0, 0), (1, 1, 1), True),                # test_minmax
             ((1.0001, 0, 1.0001), (0, 1, 0), True),      # test_edge_parallel
             ((0.99999, 0, 0.9999), (0, 1, 0), False),    # test_edge_parallel_outside
             ((1.5, 0, 1.0001), (0, 3, 0), True),         # test_face_parallel
             ((1.5, 0, 0.9999), (0, 3, 0), False)]        # test_face_parallel_outside
    for o, d, want in cases:
        for mn, mx in boxes:
            assert oracle.aabb_hit(mn, mx, o, d, 0.0, inf) is want


def test_reference_sphere_uv_test(oracle):
    """shapes.rs:200-214 test_sphere_uv (exact equality)."""
    assert oracle.sphere_uv((1, 0, 0)) == (0.5, 0.5)
    assert oracle.sphere_uv((0, 1, 0)) == (0.5, 1.0)
    assert oracle.sphere_uv((0, 0, 1)) == (0.25, 0.5)
    assert oracle.sphere_uv((-1, 0, 0)) == (0.0, 0.5)
    assert oracle.sphere_uv((0, -1, 0)) == (0.5, 0.0)
    assert oracle.sphere_uv((0, 0, -1)) == (0.75, 0.5)


def test_to_rgb_edge_cases(oracle):
    """raytrace.rs:59-68: gamma 2, clamp to 0.99999999, truncating cast, NaN -> 0."""
    assert oracle.to_rgb((0.0, 0.0, 0.0), 10) == (0, 0, 0)
    assert oracle.to_rgb((10.0, 10.0, 10.0), 10) == (255, 255, 255)
    assert oracle.to_rgb((1e9, 2.5, 0.25 * 10), 10) == (255, 127, 127)
    assert oracle.to_rgb((float("nan"), -1.0, 4.0), 16) == (0, 0, 127)


def test_sphere_hit_known_answers(oracle):
    h = oracle.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -1), 0.001, math.inf)
    assert h["t"] == 0.5 and h["p"] == [0, 0, -0.5] and h["normal"] == [0, 0, 1] and h["front_face"]
    inside = oracle.sphere_hit((0, 0, 0), 2.0, (0, 0, 0), (0, 2, 0), 0.001, math.inf)
    assert inside["t"] == 1.0 and inside["normal"] == [0, -1, 0] and not inside["front_face"]
    assert oracle.sphere_hit((0, 0, -1), 0.5, (0, 2, 0), (0, 0, -1), 0.001, math.inf) is None
    assert oracle.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -1), 0.001, 0.4) is None


# ---- scene known answers at --seed=42 (pinned there by overlaying the scene on the README renders) ----
def test_random_world_seed42(oracle):
    w = oracle.world("random", 42)
    p = w.prims()
    assert p.shape[0] == 486
    kinds = p[:, 9].astype(int)
    assert (np.sum(kinds == 0), np.sum(kinds == 1), np.sum(kinds == 2)) == (391, 73, 22)
    assert list(p[1, 1:5]) == [-10.78480397871326, 0.2, -10.315249930381933, 0.2]
    assert list(p[1, 10:13]) == [0.09918545343821861, 0.5091678367927297, 0.3013420662424381]
    assert w.probes()[0] == 0x6C72B09F0D4CF795
    info = w.bvh_info(0)
    assert (info["nodes"], info["leaves"]) == (971, 486)
    w.close()


def test_final_scene_world_seed42(oracle, earthmap):
    w = oracle.world("final_scene", 42, earthmap)
    p = w.prims()
    blocks = p[p[:, 0] == 1]
    assert blocks.shape[0] == 400
    assert list(blocks[:3, 5]) == [16.629363737053623, 17.49836163198336, 53.49750533738512]
    spheres = p[p[:, 0] == 0]
    foam = spheres[-1000:]
    assert list(foam[0, 1:4]) == [134.33933384000792, 142.57123161556723, 7.857497235638467]
    assert w.probes()[0] == 0x2E0871C80E48A980
    assert w.bvh_info(0) == {"nodes": 799, "leaves": 400, "max_depth": 9}
    assert w.bvh_info(1)["nodes"] == 1999 and w.bvh_info(1)["leaves"] == 1000
    vec, perm = w.perlin(0)
    assert np.allclose(np.linalg.norm(vec, axis=1), 1.0)
    for k in range(3):
        assert sorted(perm[k]) == list(range(1024))
    w.close()
