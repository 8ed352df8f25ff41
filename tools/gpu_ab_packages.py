"""A/B of two builds of the package on the same GPU box, frames interleaved: python tools/gpu_ab_packages.py <package dir A> <package dir B>
(a package dir holds __init__.py and lib/libmlrt_hip.so; e.g. a checkout of an older commit built in a worktree and copied next to this one).
Prints the kernel time per frame (HIP events of the context) of the bench configs, median of the interleaved repetitions."""
import importlib.util, os, sys
import numpy as np
import torch


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    A, B = load(os.path.abspath(sys.argv[1]), "mlrt_a"), load(os.path.abspath(sys.argv[2]), "mlrt_b")
    cases = [("random", 1200, 1.5, 500), ("final_scene", 800, 1.0, 1000), ("cornell_smoke", 600, 1.0, 200), ("cornell_box", 600, 1.0, 200), ("simple_light", 800, 1.5, 200), ("two_spheres", 800, 1.5, 200), ("random_chk", 800, 1.5, 200), ("earth", 800, 1.5, 200), ("debug_perlin", 600, 1.0, 200)]
    for name, W, aspect, spp in cases:
        for prec in (1, 0):
            t = {"a": [], "b": []}
            ctx = {}
            for key, m in (("a", A), ("b", B)):
                earth = m.load_earthmap() if name in ("final_scene", "earth") else None
                w = m.World(name, 42, earth)
                H = m.image_height(W, aspect)
                r = m.Renderer(w, m.default_camera(w, aspect), m.make_params(w, W, H, spp), precision=prec)
                d_rgb = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
                ctx[key] = (m, w, r, d_rgb, H)
            reps = 7 if name != "final_scene" else 4
            for i in range(reps + 2):
                for key in (("a", "b") if i % 2 == 0 else ("b", "a")):
                    m, w, r, d_rgb, H = ctx[key]
                    r.render_device(None, H, d_rgb.data_ptr(), None, None)
                    torch.cuda.synchronize()
                    if i >= 2:
                        t[key].append(r.last_kernel_ms())
            a, b = np.median(t["a"]), np.median(t["b"])
            print("%-14s prec %d: A %.3f ms  B %.3f ms  B/A %.4f   (A %s | B %s)" % (name, prec, a, b, b / a, np.round(t["a"], 2), np.round(t["b"], 2)), flush=True)
            for key in ("a", "b"):
                ctx[key][2].close(); ctx[key][1].close()


main()
