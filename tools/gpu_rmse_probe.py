import sys, os, importlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T, oracle_binding
mlrt = importlib.import_module("mu-lambda-raytracer_amd")
oracle = oracle_binding.Oracle()
earth = mlrt.load_earthmap()
for name, width, aspect, spp in (("random", 400, 1.5, 64), ("final_scene", 200, 1.0, 512), ("cornell_smoke", 160, 1.0, 128), ("earth", 160, 1.5, 128)):
    rows = np.arange(0, mlrt.image_height(width, aspect), 3)
    got, ref, info = T.render_both(mlrt, oracle, name, width, aspect, spp, mlrt.PRECISION_FP32, rows=rows, earth=earth if name in ("earth", "final_scene") else None)
    print(name, "frac %.5f rmse %.2e maxdiff %d" % T.metrics(got, ref), flush=True)
for name, width, aspect, spp in (("random_chk", 160, 1.5, 128), ("debug_perlin", 160, 1.0, 128), ("two_spheres", 160, 1.5, 128)):
    rows = np.arange(0, mlrt.image_height(width, aspect), 3)
    got, ref, info = T.render_both(mlrt, oracle, name, width, aspect, spp, mlrt.PRECISION_FP32, rows=rows, earth=None)
    print(name, "frac %.5f rmse %.2e maxdiff %d" % T.metrics(got, ref), flush=True)
