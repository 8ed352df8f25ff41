"""MLRT_TRACER_SINGLE_LIGHT: the Renderer's other RayTracer, SingleLightSourceRayTracer (raytrace.rs:110-135), through the C ABI
against the oracle's restatement (tests/test_oracle_single_light.py checks that one against the formula) on the same Philox streams.
The reference holds no output of this tracer (it is constructed in a comment only, main.rs:153): parity unpinned beyond the oracle.

Tolerances as in test_gpu_parity.py: the f64 image EQUALS the oracle's where no libm function is on the device path, at most
0.01 % of the channels differ elsewhere; the f32 mode stays under a per-channel RMSE of 1e-3."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from oracle_binding import RNG_PHILOX_SAMPLE
from test_gpu_parity import cam19, metrics, LIBM_FREE

pytestmark = pytest.mark.gpu

LIGHT, INTENSITY = (14.0, 3.0, 3.0), 1.0  # main.rs:153

WORLDS = [("simple", 200, 16.0 / 9.0, 8, LIGHT), ("random", 300, 1.5, 4, LIGHT), ("random_chk", 160, 1.5, 8, LIGHT), ("two_spheres", 160, 1.5, 8, LIGHT),
          ("simple_light", 160, 1.5, 8, (10.0, 9.0, 8.0)), ("cornell_box", 160, 1.0, 8, (278.0, 500.0, 278.0)), ("cornell_smoke", 160, 1.0, 8, (278.0, 500.0, 278.0)),
          ("earth", 160, 1.5, 8, LIGHT), ("debug_perlin", 160, 1.0, 8, LIGHT), ("final_scene", 160, 1.0, 8, (278.0, 540.0, 280.0))]


def both(mlrt, oracle, earthmap, name, width, aspect, spp, light, precision, intensity=INTENSITY, aperture=0.0, focus=None):
    earth = earthmap if name in ("earth", "final_scene") else None
    pw, ow = mlrt.World(name, 42, earth), oracle.world(name, 42, earth)
    height = mlrt.image_height(width, aspect)
    cam = mlrt.default_camera(pw, aspect, aperture, focus)
    params = mlrt.make_params(pw, width, height, spp, single_light=(light, intensity))
    ref, ref_acc = ow.render_single_light(cam19(cam), width, height, spp, light, intensity, seed=42, rng_mode=RNG_PHILOX_SAMPLE, want_accum=True)
    r = mlrt.Renderer(pw, cam, params, precision=precision)
    got = r.render()
    info = r.info()
    r.close(); ow.close(); pw.close()
    return got, ref, info


@pytest.mark.parametrize("name,width,aspect,spp,light", WORLDS)
def test_fp64_single_light_frames_equal_the_oracle(mlrt, oracle, earthmap, name, width, aspect, spp, light):
    got, ref, info = both(mlrt, oracle, earthmap, name, width, aspect, spp, light, mlrt.PRECISION_FP64)
    assert info["kernel_variant"] == "full"  # the all-features variant carries the tracer
    frac, rmse, maxdiff = metrics(got, ref)
    assert frac <= (0.0 if name in LIBM_FREE else 1e-4), (frac, rmse, maxdiff)
    assert got.max() > 0


@pytest.mark.parametrize("name,width,aspect,spp,light", [WORLDS[1], WORLDS[5], WORLDS[9]])
def test_strict_and_fp32_builds(mlrt, oracle, earthmap, name, width, aspect, spp, light):
    got, ref, _ = both(mlrt, oracle, earthmap, name, width, aspect, spp, light, mlrt.PRECISION_FP64_STRICT)
    assert metrics(got, ref)[0] <= (0.0 if name in LIBM_FREE else 1e-4)
    got, ref, _ = both(mlrt, oracle, earthmap, name, width, aspect, spp * 8, light, mlrt.PRECISION_FP32)
    assert metrics(got, ref)[1] < 1e-3


def test_lens_and_intensity(mlrt, oracle, earthmap):
    """A thin-lens camera (the lens sample is drawn as always) and another intensity."""
    got, ref, _ = both(mlrt, oracle, earthmap, "random", 240, 1.5, 6, LIGHT, mlrt.PRECISION_FP64, intensity=2.5, aperture=0.1, focus=10.0)
    assert np.array_equal(got, ref)


def test_the_tracer_needs_its_context_option_and_the_one_shot_calls_set_it(mlrt, oracle):
    w = mlrt.World("simple", 42)
    H = mlrt.image_height(160, 16.0 / 9.0)
    cam = mlrt.default_camera(w, 16.0 / 9.0)
    shaded = mlrt.make_params(w, 160, H, 4, single_light=(LIGHT, INTENSITY))
    path = mlrt.make_params(w, 160, H, 4)
    L = mlrt.lib()
    # a context created without the option runs the scene's own (sphere-only) variant: the tracer is refused, the path tracer works
    r = mlrt.Renderer(w, cam, path, precision=mlrt.PRECISION_FP64)
    assert r.info()["kernel_variant"] != "full"
    out = np.zeros((H, 160, 3), dtype=np.int32)
    rc = L.mlrt_context_render(r.ctx, C.byref(cam), C.byref(shaded), None, 0, out.ctypes.data_as(C.c_void_p), None, None)
    assert rc == mlrt.MLRT_ERR_UNSUPPORTED and b"single_light_tracer" in L.mlrt_last_error()
    want_path = r.render()
    r.close()
    # one context with the option renders both tracers; the path tracer's image does not depend on the variant
    r = mlrt.Renderer(w, cam, shaded, precision=mlrt.PRECISION_FP64, opts=mlrt.options(single_light_tracer=1))
    want = r.render()
    rc = L.mlrt_context_render(r.ctx, C.byref(cam), C.byref(path), None, 0, out.ctypes.data_as(C.c_void_p), None, None)
    assert rc == 0 and np.array_equal(out, want_path) and not np.array_equal(want, want_path)
    r.close()
    # mlrt_render / mlrt_render_multi (one-shot) create their own context with the option set
    L.mlrt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.mlrt_render(w.scene, C.byref(cam), C.byref(shaded), mlrt.PRECISION_FP64, out.ctypes.data_as(C.c_void_p), None, None) == 0, L.mlrt_last_error()
    assert np.array_equal(out, want)
    devices = (C.c_int32 * 3)(0, 0, 0)
    L.mlrt_render_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    out[:] = 0
    assert L.mlrt_render_multi(w.scene, C.byref(cam), C.byref(shaded), mlrt.PRECISION_FP64, devices, 3, out.ctypes.data_as(C.c_void_p), None, None) == 0, L.mlrt_last_error()
    assert np.array_equal(out, want)
    # bad arguments
    bad = mlrt.make_params(w, 160, H, 4, single_light=((float("nan"), 0.0, 0.0), 1.0))
    assert L.mlrt_render(w.scene, C.byref(cam), C.byref(bad), mlrt.PRECISION_FP64, out.ctypes.data_as(C.c_void_p), None, None) == mlrt.MLRT_ERR_INVALID
    bad = mlrt.make_params(w, 160, H, 4)
    bad.tracer = 7
    assert L.mlrt_render(w.scene, C.byref(cam), C.byref(bad), mlrt.PRECISION_FP64, out.ctypes.data_as(C.c_void_p), None, None) == mlrt.MLRT_ERR_INVALID
    # the command line: --tracer=single_light with the light of main.rs:153 by default
    raw = subprocess.run([mlrt.CLI_PATH, "--world=simple", "--seed=42", "--aspect_ratio=16:9", "--image_width=160", "--samples_per_pixel=4", "--format=p6",
                          "--tracer=single_light"], capture_output=True, check=True).stdout
    header = b"P6\n160 %d\n255\n" % H
    assert raw.startswith(header)
    assert np.array_equal(np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(H, 160, 3)[::-1].astype(np.int32), want)
    assert subprocess.run([mlrt.CLI_PATH, "--world=simple", "--seed=42", "--tracer=phong"], capture_output=True).returncode == 2
    w.close()
