"""Perf exploration helper (not a test): kernel time against the number of rows / samples per pixel of a frame."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mu-lambda-raytracer_amd")
import torch
def probe(name, W, aspect, prec, cases):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    dev = torch.device("cuda", 0)
    for spp, n in cases:
        r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp), precision=prec)
        rows = [int(i * H / n) for i in range(n)]
        d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
        d_rgb = torch.zeros((len(rows), W, 3), dtype=torch.int32, device=dev)
        ms = []
        for _ in range(3):
            r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, None); torch.cuda.synchronize(); ms.append(r.last_kernel_ms())
        k = min(ms[1:])
        print("prec %d %s spp %d, %d rows spread over the frame: kernel %.3f ms -> %.0f Msamples/s" % (prec, name, spp, n, k, W * n * spp / k / 1e3), flush=True)
        r.close()
    w.close()
if __name__ == "__main__":
    probe("final_scene", 800, 1.0, 0, [(1000, 800), (1000, 100), (125, 800), (250, 400), (500, 200), (1000, 8), (125, 64), (64, 125), (16, 500)])
    probe("random", 1200, 1.5, 1, [(500, 800), (500, 100), (63, 800), (125, 400), (250, 200)])
