"""Debug helper: find a (pixel, sample) whose fp32 and fp64 renders differ at spp=1-ish and print both paths."""
import importlib, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mu-lambda-raytracer_amd")
name, W, aspect = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
w = pkg.World(name, 42, earth); H = pkg.image_height(W, aspect)
cam = pkg.default_camera(w, aspect); params = pkg.make_params(w, W, H, 1)
if len(sys.argv) > 4:
    prec = pkg.PRECISION_FP64 if sys.argv[4] == "fp64" else pkg.PRECISION_FP32
    rows = np.array([int(sys.argv[5])], dtype=np.int32)
    r = pkg.Renderer(w, cam, params, precision=prec); r.render(rows); r.close()
    sys.exit(0)
imgs = {}
for prec, nm in ((pkg.PRECISION_FP64, "fp64"), (pkg.PRECISION_FP32, "fp32")):
    r = pkg.Renderer(w, cam, params, precision=prec); imgs[nm] = r.render(); r.close()
d = np.abs(imgs["fp64"] - imgs["fp32"]).max(axis=2)
ys, xs = np.nonzero(d > 16)
print("pixels differing by >16 at spp 1:", len(ys), "of", d.size)
for y, x in list(zip(ys, xs))[:3]:
    pixel = int(y) * W + int(x)
    print("=== pixel", pixel, "row", y, "col", x, imgs["fp64"][y, x], imgs["fp32"][y, x], flush=True)
    for nm in ("fp64", "fp32"):
        env = dict(os.environ, MLRT_DEBUG_PIXEL="%d,0" % pixel)
        out = subprocess.run([sys.executable, __file__, name, str(W), str(aspect), nm, str(int(y))], env=env, capture_output=True, text=True)
        print("---", nm); print("\n".join(l for l in out.stdout.split("\n") if "dbg" in l), flush=True)
