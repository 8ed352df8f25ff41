"""Does handing out a rank's rows heaviest-first shorten its step?  1/8 of the rows of a bench config on one GPU: the caller's order
against rows sorted by a measured per-row cost (each row rendered alone at low spp, kernel time) and - with the experiment of
profiles/r03_experiment_row_order.patch applied, which adds mlrt_options.row_order - against the context's own learned order (costs
from the last frame, measured inside the render kernel).  Result and why it was not kept: profiles/r03_experiment_row_order.txt.
python tools/gpu_roworder_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
rowtile = importlib.import_module("mu-lambda-raytracer_amd.rowtile")
import torch

def step_ms(r, rows, W, reps=5):
    d_rows = torch.tensor(rows, dtype=torch.int32, device="cuda")
    d_rgb = torch.zeros((len(rows), W, 3), dtype=torch.int32, device="cuda")
    for _ in range(2):
        r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def probe(name, W, aspect, spp, prec):
    earth = pkg.load_earthmap() if name == "final_scene" else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    cam = pkg.default_camera(w, aspect)
    rows = rowtile.rows_for_rank(H, 0, 8)
    have = hasattr(pkg.Options(), "row_order")
    off = pkg.options(row_order=0) if have else pkg.options()
    low = pkg.Renderer(w, cam, pkg.make_params(w, W, H, 64), precision=prec, opts=off)
    cost = []
    for j in rows:
        d_rows = torch.tensor([j], dtype=torch.int32, device="cuda")
        d_rgb = torch.zeros((1, W, 3), dtype=torch.int32, device="cuda")
        low.render_device(d_rows.data_ptr(), 1, d_rgb.data_ptr(), None, None)
        torch.cuda.synchronize()
        cost.append(low.last_kernel_ms())
    low.close()
    cost = np.array(cost)
    r = pkg.Renderer(w, cam, pkg.make_params(w, W, H, spp), precision=prec, opts=off)
    full = step_ms(r, list(range(H)), W, reps=3)
    nat = step_ms(r, rows, W)
    heavy_first = [rows[i] for i in np.argsort(-cost)]
    light_first = [rows[i] for i in np.argsort(cost)]
    hf = step_ms(r, heavy_first, W)
    lf = step_ms(r, light_first, W)
    r.close()
    if not have:
        print("prec %d %s: full frame %.2f ms; 1/8 rows: caller's order %.2f ms (%.2fx), measured-heaviest first %.2f ms (%.2fx), lightest first %.2f ms (%.2fx); row cost min/mean/max %.3f/%.3f/%.3f ms" % (
            prec, name, full, nat, full / nat, hf, full / hf, lf, full / lf, cost.min(), cost.mean(), cost.max()), flush=True)
        w.close()
        return
    r = pkg.Renderer(w, cam, pkg.make_params(w, W, H, spp), precision=prec)  # row_order = 1
    full_learned = step_ms(r, list(range(H)), W, reps=3)
    learned = step_ms(r, rows, W)
    import ctypes as C
    L = pkg.lib()
    L.mlrt_debug_row_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    perm = np.zeros(len(rows), dtype=np.int32); rays = np.zeros(len(rows), dtype=np.uint32)
    assert L.mlrt_debug_row_order(r.ctx, perm.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), len(rows)) == len(rows)
    by_caller = np.empty(len(rows)); by_caller[perm] = rays
    rank = lambda a: np.argsort(np.argsort(a))
    print("   learned cost (longest tile life) vs measured cost (ms of the row alone): Spearman %.3f; units min/mean/max %.3g/%.3g/%.3g; first 10 of the learned order: measured ms %s" % (
        np.corrcoef(rank(by_caller), rank(cost))[0, 1], by_caller.min(), by_caller.mean(), by_caller.max(), np.round(cost[perm[:10]], 2)), flush=True)
    print("prec %d %s: full frame %.2f ms (learned order %.2f ms); 1/8 rows: caller's order %.2f ms (%.2fx), measured-heaviest first %.2f ms (%.2fx), lightest first %.2f ms (%.2fx), "
          "learned order %.2f ms (%.2fx of the learned full frame); row cost min/mean/max %.3f/%.3f/%.3f ms" % (
        prec, name, full, full_learned, nat, full / nat, hf, full / hf, lf, full / lf, learned, full_learned / learned, cost.min(), cost.mean(), cost.max()), flush=True)
    r.close(); w.close()

for prec in (0, 1):
    probe("random", 1200, 1.5, 500, prec)
    probe("final_scene", 800, 1.0, 1000, prec)
