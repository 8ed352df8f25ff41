"""Perf exploration helper (not a test): do the frames of two contexts on two streams overlap (the next frame's workgroups
starting while the previous frame drains)?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mu-lambda-raytracer_amd")
import torch
def probe(name, W, aspect, spp, prec, n_rows, frames=16):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    dev = torch.device("cuda", 0)
    rows = [int(i * H / n_rows) for i in range(n_rows)]
    d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
    rs = [pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp), precision=prec) for _ in range(2)]
    outs = [torch.zeros((n_rows, W, 3), dtype=torch.int32, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    for mode in ("one stream", "two streams"):
        for timed in (False, True):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for f in range(frames):
                k = f & 1
                st = streams[k if mode == "two streams" else 0]
                rs[k].render_device(d_rows.data_ptr(), n_rows, outs[k].data_ptr(), None, st.cuda_stream)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / frames
        print("prec %d %s %d rows spp %d, %s: %.3f ms per frame" % (prec, name, n_rows, spp, mode, ms), flush=True)
    assert torch.equal(outs[0], outs[1])
    for r in rs: r.close()
    w.close()
probe("random", 1200, 1.5, 500, 1, 800)
probe("random", 1200, 1.5, 500, 1, 100)
probe("final_scene", 800, 1.0, 1000, 0, 100, frames=8)
