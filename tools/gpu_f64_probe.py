"""Perf/parity exploration helper (not a test): the product f64 build against the strict f64 build (colour sums must be
bit-equal) and its kernel time on the bench workloads."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
import torch


def sums(r, rows, W):
    d_rows = torch.tensor(rows, dtype=torch.int32, device="cuda")
    d_rgb = torch.zeros((len(rows), W, 3), dtype=torch.int32, device="cuda")
    d_acc = torch.zeros((len(rows), W, 3), dtype=torch.float64, device="cuda")
    r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), d_acc.data_ptr(), None)
    torch.cuda.synchronize()
    return d_rgb.cpu().numpy(), d_acc.cpu().numpy(), r.last_kernel_ms()


def equality(name, W, aspect, spp, n_rows=6, aperture=0.0, focus=None):
    w = pkg.World(name, 42)
    H = pkg.image_height(W, aspect)
    cam = pkg.default_camera(w, aspect, aperture, focus)
    params = pkg.make_params(w, W, H, spp)
    rows = np.unique(np.linspace(0, H - 1, n_rows).astype(np.int32))
    out = {}
    for prec, tag in ((pkg.PRECISION_FP64, "f64"), (pkg.PRECISION_FP64, "f64c"), (pkg.PRECISION_FP64_STRICT, "f64s")):
        r = pkg.Renderer(w, cam, params, precision=prec, opts=pkg.options(cull_f64=1) if tag == "f64c" else None)
        out[tag] = sums(r, rows, W)
        info = r.info()
        r.close()
        print("  %s %s: %d rows kernel %.1f ms %s" % (name, tag, len(rows), out[tag][2], info), flush=True)
    for tag in ("f64", "f64c"):
        same_rgb = np.array_equal(out[tag][0], out["f64s"][0])
        n_diff = int(np.sum(out[tag][1] != out["f64s"][1]))
        print("%s %dx%d spp %d: %s vs strict: rgb equal %s, differing f64 sums %d of %d" % (name, W, H, spp, tag, same_rgb, n_diff, out[tag][1].size), flush=True)
    w.close()


def timing(name, W, aspect, spp, prec, reps=3, **opts):
    w = pkg.World(name, 42)
    H = pkg.image_height(W, aspect)
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp), precision=prec, opts=pkg.options(**opts) if opts else None)
    d_rgb = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    ms = []
    for _ in range(reps + 1):
        r.render_device(None, H, d_rgb.data_ptr(), None, None)
        torch.cuda.synchronize()
        ms.append(r.last_kernel_ms())
    print("%s prec %d %dx%d spp %d: kernel ms %s -> %.0f Msamples/s %s" % (name, prec, W, H, spp, ["%.1f" % m for m in ms[1:]], W * H * spp / min(ms[1:]) / 1e3, r.info()), flush=True)
    r.close(); w.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["eq", "time"]
    if "eq" in which:
        equality("random", 400, 1.5, 64)
        equality("random", 1200, 1.5, 500, n_rows=3)
        equality("random", 400, 1.5, 64, aperture=0.1, focus=10.0)
        equality("final_scene", 200, 1.0, 64)
        equality("cornell_smoke", 200, 1.0, 64)
        equality("cornell_box", 200, 1.0, 64)
        equality("earth", 200, 1.5, 32)
    if "time" in which:
        timing("random", 1200, 1.5, 500, pkg.PRECISION_FP64)
        timing("random", 1200, 1.5, 500, pkg.PRECISION_FP64, cull_f64=1)
        timing("random", 1200, 1.5, 500, pkg.PRECISION_FP32)
        timing("final_scene", 800, 1.0, 200, pkg.PRECISION_FP64)
        timing("final_scene", 800, 1.0, 200, pkg.PRECISION_FP64, cull_f64=1)
        timing("final_scene", 800, 1.0, 200, pkg.PRECISION_FP32)
