"""Perf exploration helper (not a test): what one rank of an N-rank strong-scaling run of the bench workload does -
its round-robin rows on one GPU, wall time per step (render + resolve + local gather) vs the N=1 step."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
rowtile = importlib.import_module("mu-lambda-raytracer_amd.rowtile")
import torch

def probe(name, W, aspect, spp, ranks=(1, 2, 4, 8), steps=5):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    prec = int(os.environ.get('MLRT_PREC', '0'))
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp), precision=prec)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    base = None
    for n in ranks:
        rows = rowtile.rows_for_rank(H, 0, n)
        d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
        d_rgb = torch.zeros((len(rows), W, 3), dtype=torch.int32, device=dev)
        for _ in range(2):
            r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        base = base or ms
        print("prec %d %s %dx%d spp %d: 1/%d of the rows: %.2f ms/step (kernel %.2f ms) -> speedup over N=1 %.2fx (ideal %d)" % (prec, name, W, H, spp, n, ms, r.last_kernel_ms(), base / ms, n), flush=True)
    r.close(); w.close()

if __name__ == "__main__":
    ranks = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8)
    probe("random", 1200, 1.5, 500, ranks)
    probe("final_scene", 800, 1.0, 1000, ranks, steps=2)
