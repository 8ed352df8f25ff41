"""Diagnostic (not a test): where does the fp32 build differ from the oracle?  Writes gpurun_out/diffmap_*.npy"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
pkg = importlib.import_module("mu-lambda-raytracer_amd")
oracle = oracle_binding.Oracle()
name, W, aspect, spp = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
pw = pkg.World(name, 42, earth); ow = oracle.world(name, 42, earth)
H = pkg.image_height(W, aspect)
cam = pkg.default_camera(pw, aspect)
c19 = list(cam.origin) + list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) + list(cam.u) + list(cam.v) + [cam.lens_radius]
params = pkg.make_params(pw, W, H, spp)
ref, acc, stats, secs = ow.render(c19, W, H, spp, seed=42, want_accum=True)
r = pkg.Renderer(pw, cam, params); got = r.render(); r.close()
d = (got - ref)
print("oracle secs", secs, "differing", np.mean(d != 0), "rmse", np.sqrt(np.mean(d.astype(float) ** 2)) / 255, "max", np.abs(d).max())
np.save(os.path.join(ROOT, "gpurun_out", "diffmap_%s.npy" % name), np.stack([got, ref]).astype(np.uint8))
