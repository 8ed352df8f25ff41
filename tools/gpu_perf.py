"""Perf exploration helper (not a test): kernel time of the bench workloads for the library in $MLRT_LIB."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
import torch
STEPS_PER_ITER = int(os.environ.get('MLRT_STEPS', '4'))  # traversal steps per kernel-loop iteration (MLRT_STEPS in mlrt_kernel.hip)

def run(name, W, aspect, spp, reps=3, aperture=0.0, focus=None):
    earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
    w = pkg.World(name, 42, earth)
    H = pkg.image_height(W, aspect)
    cam = pkg.default_camera(w, aspect, aperture, focus)
    params = pkg.make_params(w, W, H, spp)
    r = pkg.Renderer(w, cam, params, precision=int(os.environ.get('MLRT_PREC', '0')))
    d_rgb = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    ms = []
    for _ in range(reps + 1):
        r.render_device(None, H, d_rgb.data_ptr(), None, None)
        torch.cuda.synchronize()
        ms.append(r.last_kernel_ms())
    ms = ms[1:]
    print("%-40s %-12s %4dx%-4d spp %-5d kernel ms min %.1f med %.1f -> %.0f Msamples/s  %s" % (os.path.basename(pkg.LIB_PATH), name, W, H, spp, min(ms), float(np.median(ms)),
          W * H * spp / min(ms) / 1e3, r.info()), flush=True)
    if os.environ.get('MLRT_STATS'):
        import ctypes
        st = (ctypes.c_ulonglong * 16)()
        if pkg.lib().mlrt_debug_stats(r.ctx, st) == 0:
            it, steps, sb, sl, hb, hl, samples, tiles = [int(x) for x in st][:8]
            ct, cs, ch = int(st[8]), int(st[9]), int(st[10])
            print('   primary rays: %.1f trav lane-steps/sample, %.2f specials/sample' % (int(st[11]) / samples, int(st[12]) / samples), flush=True)
            print('   shader clocks per wave-iteration: trav %.0f  special %.0f (%.0f per batch)  shade %.0f (%.0f per batch); share trav %.2f special %.2f shade %.2f' % (ct / it, cs / it, cs / max(sb, 1), ch / it, ch / max(hb, 1), ct / (ct + cs + ch), cs / (ct + cs + ch), ch / (ct + cs + ch)), flush=True)
            print('   inside a shading batch: end/scatter %.0f clocks, regeneration %.0f, new-ray setup %.0f' % (int(st[13]) / max(hb, 1), int(st[14]) / max(hb, 1), (ch - int(st[13]) - int(st[14])) / max(hb, 1)), flush=True)
            print('   starved lanes (PH_WAIT: no bank free for another tile): %.3f of all lane-iterations; tiles %d, samples/tile %.0f' % (int(st[15]) / (it * 64.0), tiles, samples / max(tiles, 1)), flush=True)
            print('   stats over %d renders: iters/sample %.2f  trav lane-steps/sample %.1f  TRAV util %.3f | special batches: %.2f lanes avg, %.2f specials/sample | shade batches: %.2f lanes avg, %.2f shades/sample' % (reps + 1, it / samples * 64 / 64, steps / samples, steps / (it * 64.0 * STEPS_PER_ITER), sl / max(sb, 1), sl / samples, hl / max(hb, 1), hl / samples), flush=True)
    r.close(); w.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c4"]
    if "c2" in which: run("random", 1200, 1.5, 500)
    if "c3" in which: run("random", 1200, 1.5, 500, aperture=0.1, focus=10.0)
    if "c4" in which: run("final_scene", 800, 1.0, 200)
    if "cornell" in which: run("cornell_box", 600, 1.0, 200)
    if "smoke" in which: run("cornell_smoke", 600, 1.0, 200)
    if "earth" in which: run("earth", 1200, 1.5, 200)
