"""Exploration helper (not a test): prints GPU-vs-oracle parity metrics and timings for a set of configurations.
Run on the GPU box:  python tools/gpu_explore.py [quick]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402

pkg = importlib.import_module("mu-lambda-raytracer_amd")
oracle = oracle_binding.Oracle()


def cam19(c):
    return list(c.origin) + list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u) + list(c.v) + [c.lens_radius]


def compare(name, width, aspect, spp, rows=None, aperture=0.0, focus=None, seed=42, precisions=("fp64", "fp32"), max_depth=50):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    pw = pkg.World(name, seed, earth)
    ow = oracle.world(name, seed, earth)
    height = pkg.image_height(width, aspect)
    cam = pkg.default_camera(pw, aspect, aperture, focus)
    params = pkg.make_params(pw, width, height, spp, max_depth)
    if rows is None:
        rows = np.arange(height, dtype=np.int32)
    rows = np.asarray(rows, dtype=np.int32)
    t0 = time.time()
    ref, acc, stats, secs = ow.render(cam19(cam), width, height, spp, max_depth=max_depth, seed=seed, rows=rows, want_accum=True)
    ref = np.stack([ref[j] for j in rows])
    for prec in precisions:
        r = pkg.Renderer(pw, cam, params, precision=pkg.PRECISION_FP64 if prec == "fp64" else pkg.PRECISION_FP32)
        t1 = time.time()
        got = r.render(rows)
        dt = time.time() - t1
        d = (got - ref).astype(np.float64)
        print("%-14s %4dx%-4d spp %-4d %s: differing %.4f%%  maxdiff %3d  rmse %.2e  | gpu %.3fs kernel %.1f ms (%.1f Msamples/s) oracle %.1fs info %s"
              % (name, width, height, spp, prec, 100 * np.mean(d != 0), int(np.abs(d).max()), np.sqrt(np.mean(d * d)) / 255.0, dt, r.last_kernel_ms(),
                 len(rows) * width * spp / max(r.last_kernel_ms(), 1e-9) / 1e3, secs, r.info()), flush=True)
        r.close()
    ow.close()
    pw.close()


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    compare("random", 400, 1.5, 8)
    compare("simple", 200, 16.0 / 9.0, 16)
    compare("cornell_box", 160, 1.0, 16)
    compare("cornell_smoke", 160, 1.0, 16)
    compare("two_spheres", 160, 1.5, 16)
    compare("simple_light", 160, 1.5, 16)
    compare("earth", 160, 1.5, 16)
    compare("debug_perlin", 160, 1.0, 16)
    compare("random_chk", 160, 1.5, 16)
    compare("final_scene", 200, 1.0, 16)
    compare("random", 400, 1.5, 8, aperture=0.1, focus=10.0)
    if not quick:
        compare("random", 1200, 1.5, 500, rows=[40, 260, 480, 700])
        compare("final_scene", 800, 1.0, 200, rows=[100, 300, 500, 700])
    # full-size timing, fp32
    pw = pkg.World("random", 42)
    cam = pkg.default_camera(pw, 1.5)
    params = pkg.make_params(pw, 1200, 800, 500)
    r = pkg.Renderer(pw, cam, params)
    for _ in range(2):
        t = time.time()
        img = r.render()
        print("C2 full render fp32: wall %.3fs kernel %.1f ms -> %.1f Msamples/s" % (time.time() - t, r.last_kernel_ms(), 480e6 / r.last_kernel_ms() / 1e3), r.info(), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "c2_fp32.npy"), img.astype(np.uint8))
