# small renders of each kernel variant under a hard timeout (used after control-flow changes to the megakernel)
for cfg in "simple 64 4" "random 96 8" "cornell_box 64 8" "final_scene 64 8" "random 400 32"; do
  set -- $cfg
  timeout -k 5 60 python - "$1" "$2" "$3" <<'PY' || { echo "FAILED/TIMEOUT: $cfg"; exit 1; }
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("mu-lambda-raytracer_amd")
name, W, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
w = pkg.World(name, 42, earth); H = pkg.image_height(W, 1.5)
r = pkg.Renderer(w, pkg.default_camera(w, 1.5), pkg.make_params(w, W, H, spp))
d = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
r.render_device(None, H, d.data_ptr(), None, None); torch.cuda.synchronize()
print(name, W, spp, "ok", r.last_kernel_ms(), "ms", int(d.sum().item()), flush=True)
PY
done
