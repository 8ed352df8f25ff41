"""The plan of the one-process multi-device path (mu-lambda-raytracer_amd/csrc/mlrt_multi.cpp: check_device_list + plan_frame, the
functions mlrt_multi_create / mlrt_multi_render follow), on the CPU through the test hook mlrt_debug_multi_plan - no device call:
what an 8-GPU node would be told to do (BASELINE.json configs[4]: final_scene 800x800 row-tiled over 8 devices), and what is
refused before any device is touched.  The rows of the frame are dealt round-robin, the analogue of the reference's one task per
row (/root/reference/src/raytrace.rs:172-186)."""
import ctypes as C
import importlib

import numpy as np
import pytest

mlrt = importlib.import_module("mu-lambda-raytracer_amd")


def plan(devices, gather, width, height):
    L = mlrt.lib()
    L.mlrt_debug_multi_plan.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    n = len(devices)
    dev = (C.c_int32 * max(1, n))(*devices)
    out = (C.c_int64 * (2 + 2 * max(1, n)))()
    rc = L.mlrt_debug_multi_plan(dev, n, gather, width, height, out)
    if rc != 0:
        return rc, None
    return 0, dict(gather=int(out[0]), stride=int(out[1]), n_rows=[int(out[2 + k]) for k in range(n)], offsets=[int(out[2 + n + k]) for k in range(n)])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("width,height", [(800, 800), (1200, 800), (400, 266), (97, 13), (64, 2)])
def test_plan_of_distinct_devices(n, width, height):
    rc, P = plan(list(range(n)), mlrt.GATHER_AUTO, width, height)
    assert rc == 0
    assert P["gather"] == (mlrt.GATHER_RCCL if n > 1 else mlrt.GATHER_COPY)
    rows = [mlrt.rows_for_device(height, k, n) for k in range(n)]
    assert P["n_rows"] == [len(r) for r in rows]
    assert sorted(j for r in rows for j in r) == list(range(height))  # every row exactly once
    assert max(P["n_rows"]) - min(P["n_rows"]) <= 1 and P["n_rows"][0] == max(P["n_rows"])  # unequal counts when n does not divide H
    assert P["stride"] == P["n_rows"][0] * width * 3  # one byte per channel
    assert P["offsets"] == [k * P["stride"] for k in range(n)]
    for k in range(n):  # a peer's rows fit its slice of the gather area
        assert P["n_rows"][k] * width * 3 <= P["stride"]


def test_configs4_on_eight_devices():
    rc, P = plan(list(range(8)), mlrt.GATHER_AUTO, 800, 800)
    assert rc == 0 and P["gather"] == mlrt.GATHER_RCCL and P["n_rows"] == [100] * 8 and P["stride"] == 240000  # 240 KB per peer


def test_repeated_devices_take_copies_and_refuse_rccl():
    rc, P = plan([0] * 8, mlrt.GATHER_AUTO, 800, 800)
    assert rc == 0 and P["gather"] == mlrt.GATHER_COPY
    rc, P = plan([0, 1, 1, 2], mlrt.GATHER_AUTO, 800, 800)
    assert rc == 0 and P["gather"] == mlrt.GATHER_COPY
    rc, _ = plan([0, 1, 1, 2], mlrt.GATHER_RCCL, 800, 800)
    assert rc == mlrt.MLRT_ERR_INVALID and b"distinct" in mlrt.lib().mlrt_last_error()
    rc, P = plan([3], mlrt.GATHER_RCCL, 800, 800)  # one device with RCCL forced: the self-transfer test of the wire
    assert rc == 0 and P["gather"] == mlrt.GATHER_RCCL


def test_bad_arguments():
    assert plan([], mlrt.GATHER_AUTO, 800, 800)[0] == mlrt.MLRT_ERR_INVALID
    assert plan(list(range(65)), mlrt.GATHER_AUTO, 800, 800)[0] == mlrt.MLRT_ERR_INVALID
    assert plan([0, -1], mlrt.GATHER_AUTO, 800, 800)[0] == mlrt.MLRT_ERR_INVALID
    assert plan([0, 1], 7, 800, 800)[0] == mlrt.MLRT_ERR_INVALID
    assert plan([0, 1], mlrt.GATHER_AUTO, 1, 800)[0] == mlrt.MLRT_ERR_INVALID


def test_create_without_a_device_fails_cleanly():
    """On a machine without a GPU mlrt_multi_create gets as far as the first context and reports MLRT_ERR_NO_DEVICE (no CPU fallback)."""
    if mlrt.device_count() > 0:
        pytest.skip("a HIP device is present")
    w = mlrt.World("random", 42)
    dev = (C.c_int32 * 8)(*range(8))
    handle = C.c_void_p()
    rc = mlrt.lib().mlrt_multi_create(w.scene, dev, 8, mlrt.PRECISION_FP64, None, mlrt.GATHER_AUTO, C.byref(handle))
    assert rc == mlrt.MLRT_ERR_NO_DEVICE and not handle.value
    w.close()
