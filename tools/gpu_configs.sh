# All BASELINE.json configs that fit one GPU at full size: kernel time and Msamples/s
python tools/gpu_perf.py c2 c3 2>&1 | grep -v amdgpu | cut -c1-140
python - <<'PY'
import importlib, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("mu-lambda-raytracer_amd")
for name, W, aspect, spp in (("final_scene", 800, 1.0, 1000), ("final_scene", 800, 1.0, 10000)):
    w = pkg.World(name, 42, pkg.load_earthmap()); H = pkg.image_height(W, aspect)
    r = pkg.Renderer(w, pkg.default_camera(w, aspect), pkg.make_params(w, W, H, spp))
    d = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    for _ in range(2):
        r.render_device(None, H, d.data_ptr(), None, None); torch.cuda.synchronize()
    ms = r.last_kernel_ms()
    print("%s %dx%d spp %d: kernel %.1f ms -> %.0f Msamples/s" % (name, W, H, spp, ms, W * H * spp / ms / 1e3), flush=True)
    r.close(); w.close()
PY
