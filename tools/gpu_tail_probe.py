"""Perf exploration helper (not a test): kernel time of every rank's rows of an N-rank split on one GPU (is the sum the N=1 frame?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mu-lambda-raytracer_amd")
rowtile = importlib.import_module("mu-lambda-raytracer_amd.rowtile")
import torch
def probe(name, W, aspect, spp, prec, n):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    dev = torch.device("cuda", 0)
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp), precision=prec)
    tot = 0.0
    for rank in range(n):
        rows = rowtile.rows_for_rank(H, rank, n)
        d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
        d_rgb = torch.zeros((len(rows), W, 3), dtype=torch.int32, device=dev)
        ms = []
        for _ in range(3):
            r.render_device(d_rows.data_ptr(), len(rows), d_rgb.data_ptr(), None, None); torch.cuda.synchronize(); ms.append(r.last_kernel_ms())
        tot += min(ms[1:])
        print("prec %d %s rank %d/%d (%d rows): kernel %.2f ms" % (prec, name, rank, n, len(rows), min(ms[1:])), flush=True)
    print("   sum over ranks %.2f ms" % tot, flush=True)
    r.close(); w.close()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
probe("final_scene", 800, 1.0, 1000, 0, n)
probe("final_scene", 800, 1.0, 1000, 0, 1)
probe("random", 1200, 1.5, 500, 1, n)
