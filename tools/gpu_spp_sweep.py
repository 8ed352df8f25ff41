"""Exploration: throughput and in-kernel stats versus samples per pixel (tile shape)."""
import importlib, os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mu-lambda-raytracer_amd")
name = sys.argv[1]
W, aspect = (1200, 1.5) if name == "random" else (800, 1.0)
earth = pkg.load_earthmap() if name == "final_scene" else None
w = pkg.World(name, 42, earth); H = pkg.image_height(W, aspect)
for spp in [int(x) for x in sys.argv[2:]]:
    rows = max(8, min(H, int(H * 400 / spp)))
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp))
    d_rows = torch.tensor(np.linspace(0, H - 1, rows).astype(np.int32), device="cuda")
    d = torch.zeros((rows, W, 3), dtype=torch.int32, device="cuda")
    for _ in range(2):
        r.render_device(d_rows.data_ptr(), rows, d.data_ptr(), None, None); torch.cuda.synchronize()
    ms = r.last_kernel_ms()
    st = (ctypes.c_ulonglong * 8)(); pkg.lib().mlrt_debug_stats(r.ctx, st)
    it, steps, sb, sl, hb, hl, samples, tiles = [int(x) for x in st]
    print("%s spp %5d rows %3d: %.1f ms -> %5.0f Msamples/s | iters/sample %.2f steps/sample %.1f TRAV util %.3f special %.1f lanes x %.2f/sample shade %.1f lanes x %.2f/sample tiles %d" % (
        name, spp, rows, ms, rows * W * spp / ms / 1e3, it / samples, steps / samples, steps / (it * 64.0), sl / max(sb, 1), sl / samples, hl / max(hb, 1), hl / samples, tiles // 2), flush=True)
    r.close()
