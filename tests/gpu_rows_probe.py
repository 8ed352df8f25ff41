"""Perf exploration helper (not a test): kernel time against the number of rows rendered (every 8th row from the bottom,
or a chosen start row): separates a launch's fixed cost from its tail."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
import torch

def probe(name, W, aspect, spp, counts=(1, 2, 4, 8, 16, 32, 64, 100), stride=8, start=0):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp))
    dev = torch.device("cuda", 0)
    for n in counts:
        rows = [(start + stride * k) % H for k in range(n)]
        d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
        d_rgb = torch.zeros((n, W, 3), dtype=torch.int32, device=dev)
        ms = []
        for _ in range(4):
            r.render_device(d_rows.data_ptr(), n, d_rgb.data_ptr(), None, None)
            torch.cuda.synchronize()
            ms.append(r.last_kernel_ms())
        print("%s %dx%d spp %d: %3d rows (from %d, stride %d): kernel %.3f ms = %.3f ms/row" % (name, W, H, spp, n, start, stride, min(ms[1:]), min(ms[1:]) / n), flush=True)
    r.close(); w.close()

if __name__ == "__main__":
    probe("random", 1200, 1.5, 500)
    probe("random", 1200, 1.5, 500, start=300, stride=1)
    probe("random", 1200, 1.5, 500, start=700, stride=1)
    probe("final_scene", 800, 1.0, 1000)
