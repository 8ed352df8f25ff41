# small renders of each kernel variant under a hard timeout (used after control-flow changes to the megakernel):
# world, width, spp[, max_depth]; spp around the tile planner's thresholds, tiny images, depth 0/1
for cfg in "simple 64 4" "random 96 8" "cornell_box 64 8" "final_scene 64 8" "random 400 32" "random 33 127" "random 33 128" "random 33 129" "random 17 500" \
           "random 2 1000" "random 64 1 0" "random 64 300 1" "cornell_smoke 48 200" "final_scene 40 640" "earth 64 64"; do
  set -- $cfg
  timeout -k 5 60 python - "$1" "$2" "$3" "${4:-50}" <<'PY' || { echo "FAILED/TIMEOUT: $cfg"; exit 1; }
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("mu-lambda-raytracer_amd")
name, W, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
earth = pkg.load_earthmap() if name in ("earth", "final_scene") else None
w = pkg.World(name, 42, earth); H = max(2, pkg.image_height(W, 1.5))
r = pkg.Renderer(w, pkg.default_camera(w, 1.5), pkg.make_params(w, W, H, spp, depth))
d = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
r.render_device(None, H, d.data_ptr(), None, None); torch.cuda.synchronize()
rows = torch.tensor([H - 1, 0], dtype=torch.int32, device="cuda"); d2 = torch.zeros((2, W, 3), dtype=torch.int32, device="cuda")
r.render_device(rows.data_ptr(), 2, d2.data_ptr(), None, None); torch.cuda.synchronize()
assert torch.equal(d2[0], d[H - 1]) and torch.equal(d2[1], d[0]), "row subset differs from the full frame"
print(name, W, spp, depth, "ok", round(r.last_kernel_ms(), 3), "ms", int(d.sum().item()), flush=True)
PY
done
