"""GPU parity tests proper: the HIP path, called through the C ABI (libmlrt_hip.so), against the CPU oracle driven by
the SAME Philox streams (oracle[Philox] vs GPU[Philox], sample for sample) on seeded inputs the oracle finishes in
seconds.

Tolerances (stated, per SURVEY.md 7 "parity ladder"):
 * fp64 parity build: the 8-bit image must EQUAL the oracle's.  Where no libm function is on the device path - the `random`,
   `simple` and `cornell_box` worlds, i.e. BASELINE configs[1] and [2]: sqrt and division are IEEE-exact on both sides and the
   samplers' cos / sin / cbrt are arithmetic of the contract (turn_cos_sin, cbrt24: the same mul / fma sequence in oracle and
   device) - not one channel may differ.  Elsewhere (sin in Checker / Noise textures, log in media, acos / atan2 in sphere_uv)
   device libm and glibc differ in the last ulp, which a deep path can amplify: at most 0.01 % of the channels.
 * fp32 fast build: per-channel RMSE < 1e-3 in [0,1] units of the 8-bit gamma-space image (BASELINE.json north_star),
   the differences being f32 rounding that flips rare accept/reject and hit/miss decisions.  Those flips are unbiased
   and independent per sample, so the RMSE falls like 1/sqrt(spp): on `random` it is ~1e-4 at every spp tested; on
   final_scene (dense overlapping foam) it is bounded by 1.0e-2 / sqrt(spp) - measured 6.7e-4 at spp 128, 2.6e-4 at 512,
   2.0e-4 at the spp 1000 of BASELINE configs[3]: below 1e-3 from spp 100 on (test_fast_build_rmse_bound_as_a_function_of_spp).
"""
import numpy as np
import pytest

from oracle_binding import RNG_PHILOX_SAMPLE

pytestmark = pytest.mark.gpu


def cam19(c):
    return list(c.origin) + list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u) + list(c.v) + [c.lens_radius]


def render_both(mlrt, oracle, name, width, aspect, spp, precision, rows=None, aperture=0.0, focus=None, seed=42, max_depth=50, earth=None):
    pw = mlrt.World(name, seed, earth)
    ow = oracle.world(name, seed, earth)
    height = mlrt.image_height(width, aspect)
    cam = mlrt.default_camera(pw, aspect, aperture, focus)
    params = mlrt.make_params(pw, width, height, spp, max_depth)
    rows = np.arange(height, dtype=np.int32) if rows is None else np.asarray(rows, dtype=np.int32)
    ref, _, _, _ = ow.render(cam19(cam), width, height, spp, max_depth=max_depth, seed=seed, rng_mode=RNG_PHILOX_SAMPLE, rows=rows)
    ref = np.stack([ref[j] for j in rows])
    r = mlrt.Renderer(pw, cam, params, precision=precision)
    got = r.render(rows)
    info = r.info()
    r.close()
    ow.close()
    pw.close()
    return got, ref, info


def metrics(got, ref):
    d = (got - ref).astype(np.float64)
    return float(np.mean(d != 0)), float(np.sqrt(np.mean(d * d)) / 255.0), int(np.abs(d).max())


SMALL = [  # world, width, aspect, spp
    ("simple", 200, 16.0 / 9.0, 16), ("random", 400, 1.5, 8), ("random_chk", 160, 1.5, 16), ("two_spheres", 160, 1.5, 16),
    ("simple_light", 160, 1.5, 16), ("cornell_box", 160, 1.0, 16), ("cornell_smoke", 160, 1.0, 16), ("earth", 160, 1.5, 16),
    ("debug_perlin", 160, 1.0, 16), ("final_scene", 200, 1.0, 16)]


LIBM_FREE = {"simple", "random", "cornell_box"}  # no sin / log / acos / atan2 on the device path of these worlds


@pytest.mark.parametrize("name,width,aspect,spp", SMALL)
def test_fp64_build_equals_oracle(mlrt, oracle, earthmap, name, width, aspect, spp):
    got, ref, info = render_both(mlrt, oracle, name, width, aspect, spp, mlrt.PRECISION_FP64, earth=earthmap if name in ("earth", "final_scene") else None)
    frac, rmse, maxdiff = metrics(got, ref)
    assert frac <= (0.0 if name in LIBM_FREE else 1e-4), (frac, rmse, maxdiff)


@pytest.mark.parametrize("name,width,aspect,spp", SMALL)
def test_fp32_build_within_tolerance(mlrt, oracle, earthmap, name, width, aspect, spp):
    # spp is raised so that the 8-bit quantisation step (1/255 = 3.9e-3) does not dominate the RMSE of a single flipped sample.
    # final_scene: its dense, overlapping foam amplifies f32 shading rounding into rare (unbiased) path flips, an error
    # that falls like 1/sqrt(spp) - 1.0e-3 at spp 128, 4e-4 at the spp 1000 of BASELINE configs[3] - so it is checked at spp 512.
    spp = spp * (32 if name == "final_scene" else 8)
    rows = np.arange(0, mlrt.image_height(width, aspect), 3)
    got, ref, info = render_both(mlrt, oracle, name, width, aspect, spp, mlrt.PRECISION_FP32, rows=rows,
                                 earth=earthmap if name in ("earth", "final_scene") else None)
    frac, rmse, maxdiff = metrics(got, ref)
    assert info["lds_scene"]
    assert rmse < 1e-3, (frac, rmse, maxdiff)


def test_baseline_config1_full_image(mlrt, oracle):
    """BASELINE.json configs[0]: --world=random --seed=42 --aspect_ratio=3:2 --image_width=400 --samples_per_pixel=8."""
    got, ref, _ = render_both(mlrt, oracle, "random", 400, 1.5, 8, mlrt.PRECISION_FP64)
    assert got.shape == (266, 400, 3)  # quirk Q13: 400 @ 3:2 -> 266 rows
    assert np.array_equal(got, ref)  # (a libm-free path: exact)
    got32, _, _ = render_both(mlrt, oracle, "random", 400, 1.5, 8, mlrt.PRECISION_FP32)
    assert metrics(got32, ref)[1] < 1e-3


def test_defocus_blur_ray_generation(mlrt, oracle):
    """BASELINE.json configs[2]: --aperture=0.1 --focus_dist=10.0 (thin-lens sampling, camera.rs:40-48)."""
    got, ref, _ = render_both(mlrt, oracle, "random", 400, 1.5, 8, mlrt.PRECISION_FP64, aperture=0.1, focus=10.0)
    assert np.array_equal(got, ref)
    plain, _, _ = render_both(mlrt, oracle, "random", 400, 1.5, 8, mlrt.PRECISION_FP64)
    assert metrics(got, plain)[1] > 0.01  # the lens changes the picture


@pytest.mark.parametrize("precision_name", ["fp64", "fp32"])
def test_baseline_config2_rows_at_full_size(mlrt, oracle, precision_name):
    """BASELINE.json configs[1] at its full size (1200x800 spp 500) on a few rows the oracle finishes in seconds."""
    precision = mlrt.PRECISION_FP64 if precision_name == "fp64" else mlrt.PRECISION_FP32
    got, ref, _ = render_both(mlrt, oracle, "random", 1200, 1.5, 500, precision, rows=[40, 480, 700])
    frac, rmse, maxdiff = metrics(got, ref)
    if precision_name == "fp64":
        assert frac == 0.0, (frac, rmse, maxdiff)  # (a libm-free path: see the module docstring)
    else:
        assert rmse < 1e-3 and maxdiff <= 3, (frac, rmse, maxdiff)


@pytest.mark.parametrize("precision_name", ["fp64", "fp32"])
def test_baseline_config3_rows_at_full_size(mlrt, oracle, precision_name):
    """BASELINE.json configs[2] (defocus blur: --aperture=0.1 --focus_dist=10.0) at its full size on a few rows: thin-lens
    ray generation (f32 in the fast build) and the lens-widened primary candidate lists against the oracle."""
    precision = mlrt.PRECISION_FP64 if precision_name == "fp64" else mlrt.PRECISION_FP32
    got, ref, _ = render_both(mlrt, oracle, "random", 1200, 1.5, 500, precision, rows=[60, 400, 640], aperture=0.1, focus=10.0)
    frac, rmse, maxdiff = metrics(got, ref)
    if precision_name == "fp64":
        assert frac == 0.0, (frac, rmse, maxdiff)
    else:
        assert rmse < 1e-3 and maxdiff <= 3, (frac, rmse, maxdiff)


@pytest.mark.parametrize("precision_name", ["fp64", "fp32"])
def test_baseline_config4_rows_at_full_size(mlrt, oracle, earthmap, precision_name):
    """BASELINE.json configs[3] at its own size and spp (final_scene 800x800 spp 1000: deep BVHs, emissive rect, media, image + noise
    textures) on three rows."""
    precision = mlrt.PRECISION_FP64 if precision_name == "fp64" else mlrt.PRECISION_FP32
    got, ref, _ = render_both(mlrt, oracle, "final_scene", 800, 1.0, 1000, precision, rows=[100, 400, 700], earth=earthmap)
    frac, rmse, maxdiff = metrics(got, ref)
    if precision_name == "fp64":
        assert frac <= 1e-4, (frac, rmse, maxdiff)
    else:
        assert rmse < 1e-3, (frac, rmse, maxdiff)


@pytest.mark.parametrize("spp", [128, 512, 1000])
def test_fast_build_rmse_bound_as_a_function_of_spp(mlrt, oracle, earthmap, spp):
    """The fast build's error against the oracle on final_scene (its worst scene) as a function of spp: RMSE <= 1.0e-2 / sqrt(spp)."""
    got, ref, _ = render_both(mlrt, oracle, "final_scene", 800, 1.0, spp, mlrt.PRECISION_FP32, rows=[60, 250, 420, 610, 760], earth=earthmap)
    frac, rmse, maxdiff = metrics(got, ref)
    print("final_scene fp32 spp %d: rmse %.2e (bound %.2e), differing channels %.3f%%, max %d" % (spp, rmse, 1.0e-2 / np.sqrt(spp), 100 * frac, maxdiff))
    assert rmse <= 1.0e-2 / np.sqrt(spp), (spp, rmse)


def test_edge_cases(mlrt, oracle):
    """max_depth 0 (raytrace.rs:87-89) gives black; max_depth 1 ends every scattered path; width not a multiple of the tile."""
    got, ref, _ = render_both(mlrt, oracle, "random", 100, 1.5, 4, mlrt.PRECISION_FP64, max_depth=0)
    assert not got.any() and not ref.any()
    got, ref, _ = render_both(mlrt, oracle, "random", 100, 1.5, 4, mlrt.PRECISION_FP64, max_depth=1)
    assert np.array_equal(got, ref)
    got, ref, _ = render_both(mlrt, oracle, "random", 131, 1.5, 3, mlrt.PRECISION_FP64, seed=7)
    assert got.shape[1] == 131 and metrics(got, ref)[0] <= 1e-4
    got, ref, _ = render_both(mlrt, oracle, "random", 64, 1.0, 5000, mlrt.PRECISION_FP64, rows=[31])  # spp > tile queue granularity
    assert np.array_equal(got, ref)
