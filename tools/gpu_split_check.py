"""Split-role kernel (mlrt_split.inc) against the one-role pair kernel: same image (f64 colour sums bit-equal), then kernel times.
   python tools/gpu_split_check.py [small|full]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mu-lambda-raytracer_amd")

def sums(world, cam, params, precision, split):
    r = pkg.Renderer(world, cam, params, precision=precision, opts=pkg.options(split_roles=split))
    H, W = params.image_height, params.image_width
    d_rgb = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    d_acc = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    ms = []
    for _ in range(3):
        r.render_device(None, H, d_rgb.data_ptr(), d_acc.data_ptr(), None)
        torch.cuda.synchronize()
        r.check()
        ms.append(r.last_kernel_ms())
    info = r.info()
    if os.environ.get("MLRT_STATS") and split:
        import ctypes
        st = (ctypes.c_ulonglong * 16)()
        if pkg.lib().mlrt_debug_stats(r.ctx, st) == 0:
            v = [int(x) for x in st]
            n = params.image_width * params.image_height * params.samples_per_pixel * 3.0
            print("   T waves: iterations/sample %.3f, lanes traversing per iteration %.1f, holding a ray %.1f, special batches %.1f lanes avg, idle iterations share %.2f" % (
                v[0] / n, v[1] / max(v[0], 1), v[5] / max(v[0], 1), v[3] / max(v[2], 1), v[4] / max(v[0], 1)))
            print("   S waves: iterations/sample %.3f, passes/sample %.4f with %.1f paths avg, generated/sample %.2f, idle share %.2f, send spins/pass %.2f" % (
                v[8] / n, v[9] / n, v[10] / max(v[9], 1), v[11] / n, v[12] / max(v[8], 1), v[13] / max(v[9], 1)))
    r.close()
    return d_rgb.cpu().numpy(), d_acc.cpu().numpy(), min(ms), info

which = sys.argv[1] if len(sys.argv) > 1 else "small"
cases = [("simple", 60, 16.0 / 9.0, 8, 0.0), ("random", 120, 1.5, 16, 0.0), ("random", 400, 1.5, 8, 0.0), ("random", 240, 1.5, 130, 0.1)]
if which == "full":
    cases = [("random", 1200, 1.5, 500, 0.0), ("random", 1200, 1.5, 500, 0.1)]
for name, W, aspect, spp, aperture in cases:
    w = pkg.World(name, 42)
    H = pkg.image_height(W, aspect)
    cam = pkg.default_camera(w, aspect, aperture, 10.0 if aperture else None)
    params = pkg.make_params(w, W, H, spp)
    for prec, pname in ((pkg.PRECISION_FP64, "f64"), (pkg.PRECISION_FP32, "f32")):
        rgb0, acc0, ms0, info0 = sums(w, cam, params, prec, 0)
        rgb1, acc1, ms1, info1 = sums(w, cam, params, prec, 1)
        print("%-8s %4dx%-4d spp %-4d aperture %.1f %s: one-role %.2f ms  split %.2f ms  colour sums differ in %d of %d, rgb in %d   %s" % (
            name, W, H, spp, aperture, pname, ms0, ms1, int((acc0 != acc1).sum()), acc0.size, int((rgb0 != rgb1).sum()), info1["kernel_variant"]), flush=True)
    w.close()
