"""The reference-precision (f64) product mode against the strict f64 build, which performs exactly the reference's box
and primitive tests (its BVH topology, bhv.rs:122-165, and AABB::hit with divisions, bhv.rs:29-42).

The product mode keeps every operation that decides a sample's value in f64 in the reference's operation order and is
free only in what it CULLS (SAH topology, box wrappers, reciprocal slab test with an outward slack, primary candidate
lists, LDS residency, kernel variant).  None of that can change a closest hit, so the claim tested here is the strongest
one available: the f64 colour SUMS of every pixel (2^-32 fixed point, order independent) are bit-identical, for every
switch of mlrt_options, not merely the 8-bit image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

WORLDS = [  # world, width, aspect, spp, aperture
    ("random", 240, 1.5, 48, 0.0), ("random", 240, 1.5, 48, 0.1), ("simple", 160, 16.0 / 9.0, 32, 0.0), ("random_chk", 160, 1.5, 32, 0.0),
    ("two_spheres", 120, 1.5, 32, 0.0), ("simple_light", 120, 1.5, 32, 0.0), ("cornell_box", 120, 1.0, 48, 0.0), ("cornell_smoke", 120, 1.0, 48, 0.0),
    ("earth", 120, 1.5, 32, 0.0), ("debug_perlin", 120, 1.0, 32, 0.0), ("final_scene", 160, 1.0, 48, 0.0)]


def colour_sums(mlrt, world, cam, params, precision, rows, opts=None):
    r = mlrt.Renderer(world, cam, params, precision=precision, opts=opts)
    d_rows = torch.tensor(rows, dtype=torch.int32, device="cuda")
    d_rgb = torch.zeros((len(rows), params.image_width, 3), dtype=torch.int32, device="cuda")
    d_acc = torch.zeros((len(rows), params.image_width, 3), dtype=torch.float64, device="cuda")
    r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), d_acc.data_ptr(), None)
    torch.cuda.synchronize()
    info = r.info()
    r.close()
    return d_rgb.cpu().numpy(), d_acc.cpu().numpy(), info


@pytest.mark.parametrize("name,width,aspect,spp,aperture", WORLDS)
def test_product_f64_colour_sums_equal_strict_f64(mlrt, earthmap, name, width, aspect, spp, aperture):
    w = mlrt.World(name, 42, earthmap if name in ("earth", "final_scene") else None)
    height = mlrt.image_height(width, aspect)
    cam = mlrt.default_camera(w, aspect, aperture, 10.0 if aperture else None)
    params = mlrt.make_params(w, width, height, spp)
    rows = np.arange(0, height, 2, dtype=np.int32)
    rgb_s, acc_s, info_s = colour_sums(mlrt, w, cam, params, mlrt.PRECISION_FP64_STRICT, rows)
    assert not info_s["lds_scene"] and info_s["block_threads"] == 256
    assert acc_s.any()
    # default: f32 culling entries (proven-outward bounds and slab test), f64 parameters in payload slots; cull_f64: f64 entries
    for opts in (None, mlrt.options(cull_f64=1)):
        rgb_p, acc_p, info_p = colour_sums(mlrt, w, cam, params, mlrt.PRECISION_FP64, rows, opts=opts)
        assert np.array_equal(rgb_p, rgb_s)
        assert np.array_equal(acc_p, acc_s), int(np.sum(acc_p != acc_s))
    w.close()


SWITCHES = [dict(sah=0), dict(leaf_wrap=0), dict(sah=0, leaf_wrap=0), dict(primary_lists=0), dict(lds_scene=0), dict(force_full=1), dict(cull_f64=1),
            dict(cull_f64=1, sah=0, leaf_wrap=0), dict(pair_walk=0), dict(pair_walk=0, primary_lists=0), dict(pair_walk=0, sah=0, leaf_wrap=0), dict(lds_scene=0, primary_lists=0),
            dict(tile_samples=64, tile_queues=1), dict(tail=[64, 64, 128], tile_queues=64)]


@pytest.mark.parametrize("name", ["random", "final_scene", "cornell_smoke"])
def test_no_option_switch_changes_an_f64_colour_sum(mlrt, earthmap, name):
    """Every field of mlrt_options (and so every MLRT_* environment variable, which only fill that struct) is result-neutral
    in the f64 product mode: bit-equal colour sums with each switch flipped."""
    w = mlrt.World(name, 42, earthmap if name == "final_scene" else None)
    cam = mlrt.default_camera(w, 1.0)
    params = mlrt.make_params(w, 96, 96, 160)
    rows = np.arange(0, 96, 3, dtype=np.int32)
    _, base, _ = colour_sums(mlrt, w, cam, params, mlrt.PRECISION_FP64, rows)
    for sw in SWITCHES:
        _, acc, info = colour_sums(mlrt, w, cam, params, mlrt.PRECISION_FP64, rows, opts=mlrt.options(**sw))
        if sw.get("lds_scene") == 0:
            assert not info["lds_scene"]
        if name == "random":  # a sphere-only world: the pair structure unless switched off (or another kernel variant / entry type is forced)
            assert info["kernel_variant"] == ("lean-pair" if not (sw.get("pair_walk") == 0 or sw.get("force_full") or sw.get("cull_f64")) else ("full" if sw.get("force_full") else "lean-threaded"))
        assert np.array_equal(acc, base), (sw, int(np.sum(acc != base)))
    w.close()


def test_fast_build_switches_keep_the_image_within_its_tolerance(mlrt):
    """The same switches in the fast (f32) build: SAH / wrappers change which f32 box tests run, not which primitives can
    be hit, so the 8-bit image moves by rare f32 culling differences only (RMSE far below the build's 1e-3 tolerance);
    tile geometry, LDS residency and the kernel variant leave it bit-identical."""
    w = mlrt.World("random", 42)
    cam = mlrt.default_camera(w, 1.5)
    params = mlrt.make_params(w, 240, 160, 64)
    r = mlrt.Renderer(w, cam, params)
    base = r.render()
    r.close()
    for sw in SWITCHES:
        r = mlrt.Renderer(w, cam, params, opts=mlrt.options(**sw))
        img = r.render()
        r.close()
        if "cull_f64" in sw:
            continue  # (an FP64 option)
        if any(k in sw for k in ("sah", "leaf_wrap", "pair_walk")):
            rmse = float(np.sqrt(np.mean((img - base).astype(np.float64) ** 2)) / 255.0)
            assert rmse < 2e-4, (sw, rmse)
        else:
            assert np.array_equal(img, base), sw
    w.close()


def test_max_depth_beyond_the_list_cursor_bits(mlrt):
    """max_depth is any i32 in the reference (main.rs:72).  The primary candidate lists keep their cursor in the upper bits of
    a lane's depth word, so they are switched off above 65535: the image must equal the walked one (and the one at depth
    50: no path of `random` gets near either bound... except that the bound is also the bounce budget, which 70000 never
    exhausts while 50 may: compare with lists off at the same depth)."""
    w = mlrt.World("random", 42)
    cam = mlrt.default_camera(w, 1.5)
    params = mlrt.make_params(w, 120, 80, 16, max_depth=70000)
    for precision in (mlrt.PRECISION_FP32, mlrt.PRECISION_FP64):
        r = mlrt.Renderer(w, cam, params, precision=precision)
        img = r.render()
        r.close()
        r = mlrt.Renderer(w, cam, params, precision=precision, opts=mlrt.options(primary_lists=0))
        walked = r.render()
        r.close()
        assert img.any() and np.array_equal(img, walked)
    w.close()


@pytest.mark.parametrize("name,width,aspect,spp,aperture", [("random", 240, 1.5, 48, 0.0), ("random", 200, 1.5, 160, 0.1), ("simple", 160, 16.0 / 9.0, 32, 0.0)])
def test_split_role_kernel_renders_the_same_sums(mlrt, name, width, aspect, spp, aperture):
    """mlrt_options.split_roles (csrc/mlrt_split.inc: traversal waves and shading waves of a workgroup handing paths to each other through
    two LDS queues; an experiment, off by default - it issues 1.8x the instructions of the one-role kernel, profiles/r03_experiment_split_roles*)
    is the same arithmetic scheduled differently: f64 colour sums bit-equal to the one-role kernel's, both precisions; a frame large
    enough for its queues to wrap many times."""
    w = mlrt.World(name, 42)
    height = mlrt.image_height(width, aspect)
    cam = mlrt.default_camera(w, aspect, aperture, 10.0 if aperture else None)
    params = mlrt.make_params(w, width, height, spp)
    rows = np.arange(height, dtype=np.int32)
    for precision in (mlrt.PRECISION_FP64, mlrt.PRECISION_FP32):
        rgb0, acc0, _ = colour_sums(mlrt, w, cam, params, precision, rows)
        rgb1, acc1, _ = colour_sums(mlrt, w, cam, params, precision, rows, opts=mlrt.options(split_roles=1))
        assert acc0.any() and np.array_equal(acc0, acc1) and np.array_equal(rgb0, rgb1)
    w.close()
