"""Prints, per world with media / libm on its device path, the fraction of 8-bit channels in which the f64 frame differs from the oracle's
(the quantity tests/test_gpu_parity.py bounds by 1e-4), at the test's size and at a larger one.  python tools/gpu_parity_frac.py [package dir]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:  # another build of the package (a directory with __init__.py and lib/), e.g. an older commit's
    import importlib.util
    path = os.path.abspath(sys.argv[1])
    spec = importlib.util.spec_from_file_location("mlrt_other", os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    pkg = importlib.util.module_from_spec(spec); sys.modules["mlrt_other"] = pkg; spec.loader.exec_module(pkg)
else:
    pkg = importlib.import_module("mu-lambda-raytracer_amd")
import oracle_binding
from test_gpu_parity import render_both, metrics

oracle = oracle_binding.Oracle()
earth = pkg.load_earthmap()
for name, width, aspect, spp in [("cornell_smoke", 160, 1.0, 16), ("final_scene", 200, 1.0, 16), ("cornell_smoke", 300, 1.0, 64), ("final_scene", 300, 1.0, 64),
                                 ("random_chk", 300, 1.5, 32), ("two_spheres", 300, 1.5, 32), ("debug_perlin", 300, 1.0, 32), ("simple_light", 300, 1.5, 32)]:
    got, ref, info = render_both(pkg, oracle, name, width, aspect, spp, pkg.PRECISION_FP64, earth=earth if name == "final_scene" else None)
    frac, rmse, maxdiff = metrics(got, ref)
    print("%-14s %4d spp %3d: %d of %d channels differ (%.2e), max |diff| %d, rmse %.2e" % (name, width, spp, int(round(frac * got.size)), got.size, frac, maxdiff, rmse), flush=True)
