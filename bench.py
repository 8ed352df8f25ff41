#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's configs, at the reference's precision.

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4|c5] [--precision fp64|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline (`value`, `dtype: "f64"`): Msamples/s of the f64 product mode (MLRT_PRECISION_FP64: every operation that decides
a sample's value in f64 in the reference's operation order, /root/reference/src/vec.rs:5-8) on configs[1] =
--world=random --seed=42 1200x800 spp 500.  The mixed-precision fast build (f32 boxes / shading, f64 ray state and
sphere quadratic) is reported next to it as the `fast` sub-record with its measured RMSE against the f64 image of the
same run - it is a mode of the product, not the number the reference is compared with.

A "step" is one full render of the image through the C ABI (mlrt_context_render_device) with the scene already resident
in HBM.  With N > 1 the image's rows are dealt to the ranks round-robin (strong scaling: the image is fixed), each rank
renders its rows on its own GPU and ONE RCCL gather (torch.distributed, backend nccl) collects the 8-bit-valued rows on
rank 0 inside the timed region (mu-lambda-raytracer_amd/rowtile.py).  PyTorch is plumbing only: device buffers,
streams, the process group.

`roofline`: the render kernel is bound by vector-instruction issue, not by HBM (the scene is LDS resident; physical HBM
traffic is ~0.1 % of peak): `bound: "valu"`, achieved = wave-level VALU instructions per second = the kernel's
SQ_INSTS_VALU per launch (rocprofv3 PMC, committed under profiles/, named in `roofline.profile`) / the kernel time
measured live with HIP events; peak = 1 wave-instruction per 2 cycles per SIMD-32 x 1024 SIMDs x 2.4 GHz
(MI355X_MICROARCH.md).  Lane utilisation and the LDS bank-conflict share ride along.  SURVEY 8(d)'s algorithmic-bytes
figure is kept as `reference_work_equivalent` (what the REFERENCE's traversal would read for the same samples, counts
from this run's cpu_baseline): it is not a utilisation and can exceed HBM peak.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # BASELINE.json configs[1..4]
    "c2": {"world": "random", "seed": 42, "aspect": 1.5, "aspect_ratio": "3:2", "image_width": 1200, "samples_per_pixel": 500, "max_depth": 50,
           "aperture": 0.0, "focus_dist": None, "baseline": "configs[1]", "metric": "Msamples/sec (pixels*spp) --world=random 1200x800 spp=500"},
    "c3": {"world": "random", "seed": 42, "aspect": 1.5, "aspect_ratio": "3:2", "image_width": 1200, "samples_per_pixel": 500, "max_depth": 50,
           "aperture": 0.1, "focus_dist": 10.0, "baseline": "configs[2]", "metric": "Msamples/sec (pixels*spp) --world=random --aperture=0.1 --focus_dist=10.0 1200x800 spp=500"},
    "c4": {"world": "final_scene", "seed": 42, "aspect": 1.0, "aspect_ratio": "1:1", "image_width": 800, "samples_per_pixel": 1000, "max_depth": 50,
           "aperture": 0.0, "focus_dist": None, "baseline": "configs[3]", "metric": "Msamples/sec (pixels*spp) --world=final_scene 800x800 spp=1000"},
    "c5": {"world": "final_scene", "seed": 42, "aspect": 1.0, "aspect_ratio": "1:1", "image_width": 800, "samples_per_pixel": 10000, "max_depth": 50,
           "aperture": 0.0, "focus_dist": None, "baseline": "configs[4]", "metric": "Msamples/sec (pixels*spp) --world=final_scene 800x800 spp=10000 row-tiled"},
}
SIMDS, CLOCK_GHZ = 1024, 2.4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, max clock 2400 MHz
VALU_PEAK_GINSTR = SIMDS * CLOCK_GHZ / 2.0  # one wave64 VALU instruction per 2 cycles per SIMD-32 (the guide's constants table)
# What this chip sustains when measured (tools/microbench/valu_issue.hip under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU,
# profiles/r03_valu_issue_microbench.txt): 4.4-4.5 cycles per plain wave64 instruction per SIMD at 2, 4 and 8 waves per SIMD, f32 and f64
# arithmetic alike (f32 transcendentals 8.4, f64 rcp / sqrt 16.2), whatever part of the wave is active.  4 cycles is the floor of that
# measurement's model (a wave64 instruction = 4 passes of 16 lanes) and what the render kernels themselves reach.
VALU_MEASURED_CYCLES = 4.0
VALU_PEAK_MEASURED_GINSTR = SIMDS * CLOCK_GHZ / VALU_MEASURED_CYCLES
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Device fp32 record sizes of SURVEY 8(d)'s algorithmic-bytes figure
REF_BYTES = {"aabb_tests": 32.0, "sphere_tests": 20.0, "rect_tests": 24.0, "scatters": 16.0, "noise_evals": 1008.0, "texel_fetches": 3.0}
# rough CPU cost of the oracle per sample per thread, to size the bounded CPU sample
CPU_US_PER_SAMPLE = {"random": 3.5, "final_scene": 6.1}


def cpu_baseline(cfg, cam19, height, seconds_budget=15.0):
    """The oracle (kind "port": a C++ restatement - the Rust reference cannot be built here) with the reference's own
    per-row PCG64 streams on a bounded sample of the same workload: evenly spaced rows, all host cores; at the spp of the
    config when a row fits the budget, else at reduced spp (samples are independent: cost is linear in spp)."""
    import numpy as np
    import oracle_binding
    oracle = oracle_binding.Oracle()
    earth = None
    if cfg["world"] in ("earth", "final_scene"):
        pkg = importlib.import_module("mu-lambda-raytracer_amd")
        earth = pkg.load_earthmap()
    ow = oracle.world(cfg["world"], cfg["seed"], earth)
    try:
        available = len(os.sched_getaffinity(0))
    except AttributeError:
        available = os.cpu_count() or 1
    # every core this process may run on (the reference's rayon pool does the same: raytrace.rs:176) - which is the smaller of the
    # affinity mask and the container's CPU quota (cgroup cpu.max): on a GPU box whose mask shows 256 hardware threads but whose share
    # is 16 cores, 256 threads run at a sixteenth of their speed each and the figure halves
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except Exception:
        quota = None
    threads = max(1, available if not quota else min(available, int(quota + 0.999)))
    w, spp = cfg["image_width"], cfg["samples_per_pixel"]
    us = CPU_US_PER_SAMPLE.get(cfg["world"], 5.0)
    sample_budget = seconds_budget * threads / (us * 1e-6)
    spp_cpu = spp
    while w * spp_cpu * threads > sample_budget and spp_cpu > 50:  # at least one row per thread
        spp_cpu //= 2
    n_rows = int(max(threads, min(height, sample_budget / (w * spp_cpu))))
    n_rows = max(threads, (n_rows // threads) * threads)
    rows = np.unique(np.linspace(0, height - 1, min(height, n_rows)).astype(np.int32))
    _, _, stats, secs = ow.render(cam19, w, height, spp_cpu, seed=cfg["seed"], rng_mode=oracle_binding.RNG_PCG_ROW, rows=rows, threads=threads)
    ow.close()
    samples = len(rows) * w * spp_cpu
    return {"value": samples / secs / 1e6, "unit": "Msamples/s", "cores": threads, "cores_available": available, "cpu_quota": quota, "kind": "port",
            "sample": "%d evenly spaced rows of the %dx%d image at spp %d%s (%.1f Msamples, %.1f s), PCG64 row streams, f64 C++ restatement of the Rust reference"
                      % (len(rows), w, height, spp_cpu, "" if spp_cpu == spp else " (config: %d; cost is linear in spp)" % spp, samples / 1e6, secs),
            "stats_per_sample": {k: round(v / stats["samples"], 3) for k, v in stats.items() if k != "samples"}}


PROFILE_ROUND = "r03"


def load_profile(config, dtype):
    """The committed PMC summary for this (config, dtype): profiles/r03_pmc_<config>_<dtype>.json (tools/gpu_pmc.sh)."""
    path = os.path.join(ROOT, "profiles", "%s_pmc_%s_%s.json" % (PROFILE_ROUND, config, dtype))
    if not os.path.exists(path):
        return None, None
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def build_stamp():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from build_stamp import stamp
        return stamp()
    except Exception:
        return None


def roofline(config, dtype, my_samples, kernel_ms, cpu_stats, spp):
    prof, prof_path = load_profile(config, dtype)
    out = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s (wave64 VALU instructions)", "frac": None, "traffic": None,
           "profile": prof_path, "kernel_ms": kernel_ms,
           "peak_note": "1 wave-instruction / 2 cycles / SIMD-32 x %d SIMDs x %.1f GHz (MI355X_MICROARCH.md); what single instruction classes sustain "
                        "against it: tools/microbench/valu_issue.hip, profiles/r03_valu_issue_microbench.txt" % (SIMDS, CLOCK_GHZ)}
    if prof:
        per_sample = prof["valu_insts_per_sample"]
        achieved = per_sample * my_samples / (kernel_ms * 1e-3) / 1e9
        out.update({"achieved": achieved, "frac": achieved / VALU_PEAK_GINSTR, "valu_insts_per_sample": per_sample,
                    "lane_utilisation": prof.get("lane_utilisation"), "lds_bank_conflict_share": prof.get("lds_bank_conflict_share"),
                    "profile_frac_at_capture": prof.get("valu_issue_frac"), "traffic": prof.get("hbm_bytes_per_launch")})
        # the same against the issue rate this chip was MEASURED to sustain (one plain wave64 instruction per 4 cycles per SIMD): at ~1.0 the
        # kernel is bound by the number of vector instructions it issues, and only fewer instructions or fuller waves make it faster
        out["peak_measured"] = VALU_PEAK_MEASURED_GINSTR
        out["frac_of_measured_peak"] = achieved / VALU_PEAK_MEASURED_GINSTR
        out["peak_measured_note"] = "profiles/r03_valu_issue_microbench.txt: 4.4-4.5 cycles per plain wave64 VALU instruction per SIMD measured (GRBM cycles / SQ_INSTS_VALU), " \
                                    "any class, any occupancy >= 2 waves, any number of active lanes; %.0f cycles taken as the peak" % VALU_MEASURED_CYCLES
        if prof.get("lane_utilisation"):
            # issue slots that carried work: frac x the share of lanes active in an issued instruction; and the lane-level instruction
            # count per sample, which falls when instructions are removed and does NOT rise when idle lanes are added
            out["useful_frac"] = out["frac"] * prof["lane_utilisation"]
            out["useful_frac_of_measured_peak"] = out["frac_of_measured_peak"] * prof["lane_utilisation"]
            out["valu_lane_insts_per_sample"] = per_sample * 64.0 * prof["lane_utilisation"]
            # the floor of THIS instruction mix: the same lane-level work issued in full waves at the measured issue rate - what perfect
            # packing of lanes would buy, and the yardstick a change that removes instructions moves
            out["full_wave_floor_ms"] = kernel_ms * out["frac_of_measured_peak"] * prof["lane_utilisation"]
        # the committed counters describe the build they were captured on: compare with the library about to be timed
        # (hipcc's output is not bit-reproducible - two compiles of one source give two hashes - so the SOURCES decide staleness and the
        # binary's hash is reported next to it: equal = the very file the counters were read from)
        here, there = build_stamp(), prof.get("build_stamp")
        out["profile_build"] = there
        out["profile_stale"] = (not here) or (not there) or here.get("source_sha256") != there.get("source_sha256")
        out["profile_same_binary"] = bool(here and there and here.get("lib_sha256") == there.get("lib_sha256"))
        if prof.get("hbm_bytes_per_launch"):
            out["hbm_physical_GBps"] = prof["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
            out["hbm_physical_frac"] = out["hbm_physical_GBps"] / HBM_PEAK_GBS
    if cpu_stats:
        b = sum(REF_BYTES[k] * cpu_stats.get(k, 0.0) for k in REF_BYTES) + 12.0 / spp
        out["reference_work_equivalent"] = {
            "bytes_per_sample": b, "GBps": b * my_samples / (kernel_ms * 1e-3) / 1e9,
            "note": "SURVEY 8(d): bytes the REFERENCE traversal (its BVH topology, counts from this run's cpu_baseline.stats_per_sample) would read "
                    "for the same samples in the device's fp32 record sizes; not a utilisation of anything - the device BVH is SAH-built and LDS resident"}
    return out


def end_to_end(pkg, torch, cfg, world, cam, params, precision, dev, W, H, spp):
    """The reference's own timing window (/root/reference/src/main.rs:145-174: renderer construction + render()), which for the device path
    is context creation (validate + SAH + pack + upload), the first frame and the download of the image - measured once each, outside the
    timed region - and the same window as the CLI reports it (`mlrt ... > /dev/null`: "Rendered in X s" on stderr, process start excluded)."""
    import subprocess
    rec = {}
    try:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        renderer = pkg.Renderer(world, cam, params, precision=precision, device=dev.index or 0)
        t1 = time.perf_counter()
        d_rgb = torch.zeros((H, W, 3), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        renderer.render_device(None, H, d_rgb.data_ptr(), None, None)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        host = d_rgb.cpu()
        t4 = time.perf_counter()
        renderer.close()
        del host
        rec.update({"context_create_ms": (t1 - t0) * 1e3, "first_frame_ms": (t3 - t2) * 1e3, "d2h_ms": (t4 - t3) * 1e3,
                    "window_ms": (t1 - t0 + t3 - t2 + t4 - t3) * 1e3, "window_Msamples_per_s": W * H * spp / (t1 - t0 + t3 - t2 + t4 - t3) / 1e6})
        # the blocking host-buffer form of the ABI (mlrt_context_render: allocations, kernels, download), second call = steady state
        renderer = pkg.Renderer(world, cam, params, precision=precision, device=dev.index or 0)
        renderer.render()
        t5 = time.perf_counter()
        renderer.render()
        rec["host_render_ms"] = (time.perf_counter() - t5) * 1e3
        renderer.close()
    except Exception as e:  # noqa: BLE001 - a diagnostic record must not take the bench line down
        rec["error"] = repr(e)
    try:
        cli = [pkg.CLI_PATH, "--world=" + cfg["world"], "--seed=%d" % cfg["seed"], "--aspect_ratio=" + cfg["aspect_ratio"], "--image_width=%d" % W,
               "--samples_per_pixel=%d" % spp, "--precision=" + ("fp64" if precision == pkg.PRECISION_FP64 else "fp32"), "--format=p6"]
        if cfg["aperture"]:
            cli += ["--aperture=%g" % cfg["aperture"], "--focus_dist=%g" % cfg["focus_dist"]]
        if cfg["world"] in ("earth", "final_scene"):
            cli += ["--earthmap=" + pkg.EARTHMAP_PPM]
        t0 = time.perf_counter()
        r = subprocess.run(cli, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
        wall = time.perf_counter() - t0
        line = [ln for ln in r.stderr.decode(errors="replace").splitlines() if "Rendered in" in ln]
        if r.returncode == 0 and line:
            secs = float(line[-1].split("Rendered in")[1].strip().rstrip("s"))
            ready = [ln for ln in r.stderr.decode(errors="replace").splitlines() if "Device ready in" in ln]
            # "Device ready in": the process's first GPU operation (HIP runtime and queues coming up), which the CLI does before it starts
            # the reference's timer; process_wall_s has everything
            rec["cli"] = {"rendered_in_s": secs, "Msamples_per_s": W * H * spp / secs / 1e6,
                          "device_ready_s": float(ready[-1].split("Device ready in")[1].strip().rstrip("s")) if ready else None, "process_wall_s": wall, "command": " ".join(os.path.basename(c) if i == 0 else c for i, c in enumerate(cli))}
        else:
            rec["cli"] = {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-200:])}
    except Exception as e:  # noqa: BLE001
        rec["cli"] = {"error": repr(e)}
    return rec


def main_inproc(args, cfg):
    """--multi inproc: the product's own multi-device path (mlrt_multi_create / mlrt_multi_render, csrc/mlrt_multi.cpp) from ONE process:
    rows dealt to the devices round-robin, one RCCL gather to devices[0], the image returned in host memory.  A step = one
    mlrt_multi_render call (so, unlike the per-rank mode, the download of the image is inside the timed region)."""
    import numpy as np
    pkg = importlib.import_module("mu-lambda-raytracer_amd")
    n_dev = pkg.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    if args.gpus > n_dev:
        raise SystemExit("bench.py --multi inproc --gpus %d: only %d device(s) visible" % (args.gpus, n_dev))
    W, spp = cfg["image_width"], cfg["samples_per_pixel"]
    H = pkg.image_height(W, cfg["aspect"])
    world = pkg.World(cfg["world"], cfg["seed"])
    cam = pkg.default_camera(world, cfg["aspect"], cfg["aperture"], cfg["focus_dist"])
    params = pkg.make_params(world, W, H, spp, cfg["max_depth"])
    prec = pkg.PRECISION_FP64 if args.precision == "fp64" else pkg.PRECISION_FP32
    multi = pkg.MultiRenderer(world, cam, params, list(range(args.gpus)), precision=prec)
    for _ in range(args.warmup):
        multi.render()
    frame_ms, kernel_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        multi.render()
        info = multi.info()
        frame_ms.append(info["frame_ms"]); kernel_ms.append(info["kernel_ms"])
    elapsed = time.perf_counter() - t0
    multi.close()
    world.close()
    value = W * H * spp * args.steps / elapsed / 1e6
    out = {"metric": cfg["metric"], "value": value, "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64" if args.precision == "fp64" else "f32", "data": "synthetic",
           "config": {"workload": "--world=%s --seed=%d --aspect_ratio=%s --image_width=%d --samples_per_pixel=%d (BASELINE.json %s)"
                                  % (cfg["world"], cfg["seed"], cfg["aspect_ratio"], W, spp, cfg["baseline"]),
                      "parallelism": "inproc: mlrt_multi_render over devices 0..%d of one process, rows round-robin, one gather (%s), image returned in host memory"
                                     % (args.gpus - 1, {1: "RCCL", 2: "hipMemcpyPeer"}.get(info["gather"], "?")),
                      "frame_ms_mean": float(np.mean(frame_ms)), "kernel_ms_per_device_mean": [float(x) for x in np.mean(np.array(kernel_ms), axis=0)]}}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default="fp64", choices=["fp32", "fp64"], help="the headline mode (default: the reference's precision)")
    ap.add_argument("--no-fast", action="store_true", help="skip the fast (f32) sub-record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi", default="ranks", choices=["ranks", "inproc"],
                    help="ranks (default): one process per GPU (torch.distributed.run); inproc: ONE process drives --gpus devices through the C ABI's "
                         "mlrt_multi_render (mlrt_multi.cpp: a context and a stream per device, one RCCL gather) - what a Rust / CLI caller gets")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end sub-record (context creation, first frame, image download, the CLI's window)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (rows gathered through host memory) only to rehearse N>1 on a box with fewer GPUs than ranks")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.multi == "inproc":
        if world_size != 1:
            raise SystemExit("bench.py --multi inproc is ONE process for all --gpus devices: start it without torch.distributed.run")
        return main_inproc(args, cfg)
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world_size))
    pkg = importlib.import_module("mu-lambda-raytracer_amd")
    rowtile = importlib.import_module("mu-lambda-raytracer_amd.rowtile")
    if not os.path.exists(pkg.LIB_PATH):
        importlib.import_module("mu-lambda-raytracer_amd.build").build()
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs are visible (one rank per GPU)" % (local_rank, n_dev))
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")

    W, spp = cfg["image_width"], cfg["samples_per_pixel"]
    H = pkg.image_height(W, cfg["aspect"])
    world = pkg.World(cfg["world"], cfg["seed"])
    cam = pkg.default_camera(world, cfg["aspect"], cfg["aperture"], cfg["focus_dist"])
    params = pkg.make_params(world, W, H, spp, cfg["max_depth"])
    my_rows = rowtile.rows_for_rank(H, rank, world_size)
    d_rows = torch.tensor(my_rows, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    gather_device = torch.device("cpu") if (args.backend == "gloo" and world_size > 1) else dev  # gloo: rehearsal only
    gatherer = rowtile.RowGatherer(H, W, rank, world_size, gather_device)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(precision, steps, warmup):
        """W untimed + K timed steps of one precision mode; returns (seconds [max over ranks], per-step HIP-event ms, last kernel ms, info, frame)."""
        renderer = pkg.Renderer(world, cam, params, precision=precision, device=device_index)
        d_rgb = torch.zeros((len(my_rows), W, 3), dtype=torch.int32, device=dev)
        frame = {}

        def step():
            renderer.render_device(d_rows.data_ptr(), len(my_rows), d_rgb.data_ptr(), None, stream.cuda_stream)
            # the one collective of the path: finished rows to rank 0 over xGMI (a plain scatter into the frame at N = 1)
            frame["image"] = gatherer.gather(d_rgb if gather_device == dev else d_rgb.cpu())

        for _ in range(warmup):
            step()
        barrier()
        # HIP events on the launch stream around every launch (candidate pre-pass + render kernel + resolve kernel), before the gather
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def timed_step(k):
            ev[k][0].record(stream)
            renderer.render_device(d_rows.data_ptr(), len(my_rows), d_rgb.data_ptr(), None, stream.cuda_stream)
            ev[k][1].record(stream)
            frame["image"] = gatherer.gather(d_rgb if gather_device == dev else d_rgb.cpu())

        t0 = time.perf_counter()
        for k in range(steps):
            timed_step(k)
        barrier()
        elapsed = time.perf_counter() - t0
        if world_size > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        step_ms = [a.elapsed_time(b) for a, b in ev]
        kernel_ms = float(sum(step_ms) / len(step_ms))  # average launch duration over the timed region
        info = renderer.info()
        image = None if frame["image"] is None else frame["image"].clone()
        renderer.close()
        return elapsed, step_ms, kernel_ms, info, image

    head_prec = pkg.PRECISION_FP64 if args.precision == "fp64" else pkg.PRECISION_FP32
    head_dtype = "f64" if args.precision == "fp64" else "f32"
    elapsed, step_ms, kernel_ms, info, image = measure(head_prec, args.steps, args.warmup)
    fast = None
    if args.precision == "fp64" and not args.no_fast:
        f_steps = max(1, min(args.steps, 50))
        f_elapsed, f_step_ms, f_kernel_ms, f_info, f_image = measure(pkg.PRECISION_FP32, f_steps, min(args.warmup, 2))
        if rank == 0:
            d = (f_image.to(torch.float64) - image.to(torch.float64))
            fast = {"dtype": "f32", "value": W * H * spp * f_steps / f_elapsed / 1e6, "unit": "Msamples/s", "steps": f_steps, "ms_per_step": f_elapsed / f_steps * 1e3,
                    "launch_ms_hip_events_mean": f_kernel_ms, "rmse_vs_f64_image": float(torch.sqrt(torch.mean(d * d)).item() / 255.0),
                    "differing_channels": float((d != 0).double().mean().item()), "max_abs_diff_8bit": int(d.abs().max().item()),
                    "mode": "MLRT_PRECISION_FP32: f32 boxes / rects / shading / primary rays, f64 ray state and sphere quadratic; a mode of the product, "
                            "below the reference's precision (tolerance: per-channel RMSE < 1e-3 against the f64 image)", **f_info}

    e2e = None
    if rank == 0 and world_size == 1 and not args.no_end_to_end:
        e2e = end_to_end(pkg, torch, cfg, world, cam, params, head_prec, dev, W, H, spp)
    if rank == 0:
        total_samples = W * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        my_samples = len(my_rows) * W * spp
        out = {
            "metric": cfg["metric"], "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": head_dtype, "data": "synthetic",
            "config": {"workload": "--world=%s --seed=%d --aspect_ratio=%s --image_width=%d --samples_per_pixel=%d%s (BASELINE.json %s)"
                                   % (cfg["world"], cfg["seed"], cfg["aspect_ratio"], W, spp,
                                      "" if not cfg["aperture"] else " --aperture=%g --focus_dist=%g" % (cfg["aperture"], cfg["focus_dist"]), cfg["baseline"]),
                       "precision_mode": "MLRT_PRECISION_FP64 (f64 arithmetic in the reference's operation order; device-built SAH culling structure)" if head_dtype == "f64" else "MLRT_PRECISION_FP32",
                       "rows_per_rank": len(my_rows), "row_tiling": "round-robin rows, one gather", "rng": "philox4x32-10 per (pixel, sample)",
                       "launch_ms_hip_events_mean": kernel_ms, "launch_ms_hip_events_median": float(np.median(step_ms)),
                       "launch_ms_hip_events_min_max": [float(min(step_ms)), float(max(step_ms))], **info},
        }
        cpu = None
        if world_size == 1 and not args.no_cpu_baseline:
            cam19 = list(cam.origin) + list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) + list(cam.u) + list(cam.v) + [cam.lens_radius]
            cpu = cpu_baseline(cfg, cam19, H)
        out["roofline"] = roofline(args.config, head_dtype, my_samples, kernel_ms, cpu["stats_per_sample"] if cpu else None, spp)
        if fast:
            fr = roofline(args.config, "f32", my_samples, fast["launch_ms_hip_events_mean"], None, spp)
            fast["roofline"] = {k: fr[k] for k in ("bound", "achieved", "peak", "unit", "frac", "lane_utilisation", "profile") if k in fr}
            out["fast"] = fast
        if cpu:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]  # (vs_baseline stays null: BASELINE.md holds no published number for this metric)
        if e2e:
            out["end_to_end"] = e2e
        print(json.dumps(out), flush=True)
    world.close()
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
