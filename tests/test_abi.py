"""The C-ABI library loads on a machine without a GPU, exports every symbol the headers declare, and its compute
entry points fail loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlrt_[a-z0-9_]+)\s*\(", text)) - {"mlrt_progress_fn"})


def test_every_declared_symbol_is_exported(mlrt):
    L = mlrt.lib()
    for header, listed in (("mlrt.h", mlrt.EXPORTS_MLRT_H), ("mlrt_host.h", mlrt.EXPORTS_MLRT_HOST_H)):
        declared = _declared(header)
        assert declared == sorted(listed), header
        for name in declared:
            assert hasattr(L, name), name


def test_version(mlrt):
    major, minor = C.c_int32(-1), C.c_int32(-1)
    mlrt.lib().mlrt_version(C.byref(major), C.byref(minor))
    assert (major.value, minor.value) == (0, 1)


def test_struct_sizes_match_the_header(mlrt):
    # natural-alignment sizes of the POD structs in include/mlrt.h
    assert C.sizeof(mlrt.Material) == 40 and C.sizeof(mlrt.Texture) == 48 and C.sizeof(mlrt.Sphere) == 40 and C.sizeof(mlrt.Box) == 56
    assert C.sizeof(mlrt.Rect) == 56 and C.sizeof(mlrt.BvhNode) == 56 and C.sizeof(mlrt.Transform) == 56 and C.sizeof(mlrt.Object) == 32
    assert C.sizeof(mlrt.CameraStruct) == 152 and C.sizeof(mlrt.Params) == 112


def test_invalid_scene_is_rejected_before_touching_a_device(mlrt):
    w = mlrt.World("random", 42)
    scene = mlrt.Scene()
    C.memmove(C.byref(scene), w.scene, C.sizeof(mlrt.Scene))
    scene.n_objects = 0
    ctx = C.c_void_p()
    rc = mlrt.lib().mlrt_context_create(C.byref(scene), 0, mlrt.PRECISION_FP32, C.byref(ctx))
    assert rc == mlrt.MLRT_ERR_INVALID and not ctx.value
    assert b"no objects" in mlrt.lib().mlrt_last_error()
    w.close()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_a_device(mlrt):
    assert mlrt.device_count() == 0
    w = mlrt.World("random", 42)
    cam = mlrt.default_camera(w, 1.5)
    p = mlrt.make_params(w, 40, 26, 2)
    with pytest.raises(mlrt.MlrtError) as e:
        mlrt.Renderer(w, cam, p)
    assert e.value.code == mlrt.MLRT_ERR_NO_DEVICE
    w.close()


def test_headers_are_valid_c99(tmp_path):
    """include/*.h must be consumable from plain C (the FFI layer of a Rust/Go host sees exactly this)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c_abi_consumer.c")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], check=True)


def test_row_plan_of_the_multi_device_path(mlrt):
    """mlrt_rows_for_device (the C-side row plan of mlrt_multi_*): every row to exactly one device, round-robin, the same
    plan as rowtile.rows_for_rank (the one-process-per-GPU path of bench.py); device 0 holds the most rows."""
    import importlib
    rowtile = importlib.import_module("mu-lambda-raytracer_amd.rowtile")
    for height in (1, 2, 7, 266, 800):
        for n in (1, 2, 3, 4, 8):
            seen = []
            for k in range(n):
                rows = mlrt.rows_for_device(height, k, n)
                assert rows == rowtile.rows_for_rank(height, k, n)
                assert len(rows) <= len(mlrt.rows_for_device(height, 0, n))
                assert mlrt.lib().mlrt_rows_for_device(height, k, n, None) == len(rows)
                seen += rows
            assert sorted(seen) == list(range(height))
    assert mlrt.lib().mlrt_rows_for_device(10, 3, 3, None) == 0 and mlrt.lib().mlrt_rows_for_device(10, -1, 3, None) == 0


def test_multi_device_entry_points_fail_loudly_without_a_device_or_with_bad_arguments(mlrt):
    w = mlrt.World("random", 42)
    handle = C.c_void_p()
    dev = (C.c_int32 * 2)(0, 0)
    rc = mlrt.lib().mlrt_multi_create(w.scene, dev, 0, mlrt.PRECISION_FP64, None, mlrt.GATHER_AUTO, C.byref(handle))
    assert rc == mlrt.MLRT_ERR_INVALID and not handle.value
    rc = mlrt.lib().mlrt_multi_create(w.scene, dev, 2, mlrt.PRECISION_FP64, None, mlrt.GATHER_RCCL, C.byref(handle))
    assert rc == mlrt.MLRT_ERR_INVALID and b"distinct" in mlrt.lib().mlrt_last_error()
    if not os.path.exists("/dev/kfd"):
        rc = mlrt.lib().mlrt_multi_create(w.scene, dev, 1, mlrt.PRECISION_FP64, None, mlrt.GATHER_AUTO, C.byref(handle))
        assert rc == mlrt.MLRT_ERR_NO_DEVICE and not handle.value
    w.close()


def test_options_defaults_and_environment(mlrt, monkeypatch):
    o = mlrt.Options()
    mlrt.lib().mlrt_options_default(C.byref(o))
    assert (o.sah, o.leaf_wrap, o.primary_lists, o.lds_scene, o.force_full, o.cull_f64) == (1, 1, 1, 1, 0, 0)
    assert (o.tile_samples, o.tile_queues, o.n_tail, o.stats, o.debug_pixel) == (512, 8, -1, 0, -1)
    monkeypatch.setenv("MLRT_NO_SAH", "1")
    monkeypatch.setenv("MLRT_TILE_TAIL", "64,128")
    monkeypatch.setenv("MLRT_CULL_F64", "1")
    mlrt.lib().mlrt_options_from_env(C.byref(o))
    assert o.sah == 0 and o.cull_f64 == 1 and o.n_tail == 2 and list(o.tail)[:2] == [64, 128]


def test_malformed_scenes_are_refused_with_a_message(mlrt):
    """validate() (csrc/mlrt_api.cpp) runs before any device call: every malformed field below must come back as MLRT_ERR_INVALID
    with a message naming it - never a crash, a hang or a silently different picture."""
    import directed_cases as dc

    def rc_of(mutate):
        sc = dc.scene_bvh(mlrt)
        scene = sc.to_ctypes()
        mutate(scene)
        ctx = C.c_void_p()
        rc = mlrt.lib().mlrt_context_create(C.byref(scene), 0, mlrt.PRECISION_FP64, C.byref(ctx))
        assert not ctx.value
        return rc, mlrt.lib().mlrt_last_error().decode()

    def nan_sphere(s): s.spheres[3].center[1] = float("nan")
    def inf_radius(s): s.spheres[0].radius = float("inf")
    def nan_box(s): s.boxes[2].max[0] = float("nan")
    def nan_node(s): s.bvh_nodes[1].bounds_min[2] = float("nan")
    def nan_offset(s): s.transforms[0].offset[0] = float("inf")
    def null_array(s): s.spheres = C.cast(None, type(s.spheres))
    def bad_child(s): s.bvh_nodes[0].left = s.n_bvh_nodes
    def cyclic(s): s.bvh_nodes[s.bvhs[0].root].left = s.bvhs[0].root
    def bad_material(s): s.boxes[0].material = s.n_materials
    def bad_texture(s): s.materials[0].texture = -1
    def self_checker(s): s.textures[4].index = 4  # base(): texture 4 is the Checker
    def bad_perm(s): C.cast(s.perlins[0].perm, C.POINTER(C.c_int32))[5] = 1024
    def bad_axis(s): s.transforms[0].axis = 3
    def bad_object(s): s.objects[0].kind = 7
    def reversed_box(s): s.boxes[1].min[1] = s.boxes[1].max[1] + 1.0

    cases = [(nan_sphere, "sphere"), (inf_radius, "sphere"), (nan_box, "box"), (nan_node, "bvh node bounds"), (nan_offset, "transform"), (null_array, "NULL"),
             (bad_child, "bvh child"), (cyclic, "cyclic"), (bad_material, "box material"), (bad_texture, "material texture"), (self_checker, "checker"),
             (bad_perm, "perlin permutation"), (bad_axis, "rotation axis"), (bad_object, "object kind"), (reversed_box, "box min")]
    for mutate, word in cases:
        rc, msg = rc_of(mutate)
        assert rc == mlrt.MLRT_ERR_INVALID, (mutate.__name__, rc, msg)
        assert word in msg, (mutate.__name__, msg)
