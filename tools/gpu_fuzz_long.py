"""Bug hunt (not a test): the randomised parity checks of tests/test_gpu_fuzz.py over many more seeds.  Prints every disagreement.
usage: python tools/gpu_fuzz_long.py [first_seed] [n_seeds]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests")); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, torch  # noqa: F401
import test_gpu_fuzz as fz, directed_cases as dc, oracle_binding as ob, test_gpu_directed as td
import importlib
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
m = td.dbg.__wrapped__()          # debug library: the probes
rel = importlib.import_module("mu-lambda-raytracer_amd")  # release library: whole frames
orc = ob.Oracle()
t0 = time.time(); bad_total = 0
for seed in range(first, first + count):
    place = sorted(fz.PLACES)[seed % 4]
    sc, lo, hi = fz.random_scene(m, seed, *fz.PLACES[place])
    scene = sc.to_ctypes()
    rays, psd = fz.random_rays(seed, lo, hi, fz.N_RAYS)
    o = ob.OracleScene(orc, scene); e = o.probe_rays(dc.SEED + seed, rays, psd); o.close()
    for build in ("fp64", "fp64_strict") + (("fp32",) if place == "unit" else ()):
        ctx = td.context(m, scene, build)
        g = np.zeros((fz.N_RAYS, 24))
        td.check(m, m.lib().mlrt_debug_probe_rays(ctx, dc.SEED + seed, fz.N_RAYS, td.ptr(rays), td.ptr(psd), td.ptr(g)))
        m.lib().mlrt_context_destroy(ctx)
        flags = [0, 10, 11, 21, 22]
        same = np.all(g[:, flags] == e[:, flags], axis=1)
        hit = (e[:, 0] == 1) & same
        tol = 1e-12 if build != "fp32" else 3e-5
        scale = np.maximum(1.0, np.abs(e[:, 1:5]).max(axis=1))
        close = ~hit | ((np.abs(g[:, 1] - e[:, 1]) <= tol * np.maximum(1.0, np.abs(e[:, 1]))) & (np.abs(g[:, 2:8] - e[:, 2:8]).max(axis=1) <= tol * scale) &
                        (np.abs(g[:, 12:21] - e[:, 12:21]).max(axis=1) <= tol * np.maximum(scale, np.abs(e[:, 12:21]).max(axis=1))))
        bad = ~(same & close)
        if build == "fp32":
            if bad.sum() > 12: print("seed", seed, place, build, "soft mismatches", int(bad.sum()), flush=True); bad_total += 1
        elif bad.any():
            i = int(np.flatnonzero(bad)[0]); bad_total += 1
            print("seed", seed, place, build, int(bad.sum()), "rays differ; first", i, rays[i].tolist(), "\n   g", g[i].tolist(), "\n   e", e[i].tolist(), flush=True)
    # whole frames, release library
    if place == "unit":
        rng = np.random.default_rng(500 + seed)
        w, h, spp = 48 + 8 * (seed % 5), 40, [8, 33, 64, 130][seed % 4]
        cam = fz.random_camera(rel, rng, lo, hi, w / h)
        sc2, _, _ = fz.random_scene(rel, seed, *fz.PLACES[place])
        for pname, prec in (("fp64", rel.PRECISION_FP64), ("fp32", rel.PRECISION_FP32)):
            got, ref, _ = fz.render_scene_both(rel, orc, sc2.to_ctypes(), cam, w, h, spp, [50, 5, 1, 12][seed % 4], 1000 + seed, int(seed % 4 != 0), prec)
            frac, rmse = fz.diff(got, ref)
            if (pname == "fp64" and frac > 5e-4) or (pname == "fp32" and rmse > 1.0e-2 / np.sqrt(spp) + 2e-4):
                print("seed", seed, "FULL frame", pname, "frac %.2e rmse %.2e" % (frac, rmse), flush=True); bad_total += 1
    sc3, lo3, hi3 = fz.sphere_scene(rel, seed)
    rng = np.random.default_rng(700 + seed)
    w, h = [(96, 64), (33, 17), (128, 40), (60, 60)][seed % 4]; spp = [16, 127, 64, 200][seed % 4]
    cam = fz.random_camera(rel, rng, lo3, hi3, w / h, above_ground=True)
    scene3 = sc3.to_ctypes()
    got, ref, _ = fz.render_scene_both(rel, orc, scene3, cam, w, h, spp, 50 if seed % 5 else 3, 2000 + seed, 1, rel.PRECISION_FP64)
    frac, rmse = fz.diff(got, ref)
    if frac > 5e-4: print("seed", seed, "LEAN frame fp64 frac %.2e rmse %.2e" % (frac, rmse), flush=True); bad_total += 1
    other, _, _ = fz.render_scene_both(rel, orc, scene3, cam, w, h, spp, 50 if seed % 5 else 3, 2000 + seed, 1, rel.PRECISION_FP64, opts=rel.options(primary_lists=0))
    if not np.array_equal(other, got): print("seed", seed, "LEAN frame: candidate lists change the image", flush=True); bad_total += 1
    # round 3: the same frame over the threaded list instead of the pair structure, and through the split-role kernel
    for label, o2 in (("threaded list", rel.options(pair_walk=0)), ("split roles", rel.options(split_roles=1))):
        other, _, _ = fz.render_scene_both(rel, orc, scene3, cam, w, h, spp, 50 if seed % 5 else 3, 2000 + seed, 1, rel.PRECISION_FP64, opts=o2)
        if not np.array_equal(other, got): print("seed", seed, "LEAN frame: %s changes the image" % label, flush=True); bad_total += 1
    if seed % 10 == 9: print("... through seed", seed, "%.0f s" % (time.time() - t0), "disagreements so far:", bad_total, flush=True)
print("done:", count, "seeds,", bad_total, "disagreements")
