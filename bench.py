#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's config:
Msamples/s (pixels*spp) for --world=random --seed=42 1200x800 spp=500 (configs[1], the book-1 cover scene).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full render of the image through the C ABI (mlrt_context_render_device) with the scene already
resident in HBM.  With N > 1 the image's rows are dealt to the ranks round-robin (strong scaling: the image is
fixed), each rank renders its rows on its own GPU and ONE RCCL gather (torch.distributed, backend nccl) collects the
8-bit-valued rows on rank 0 inside the timed region (mu-lambda-raytracer_amd/rowtile.py).  PyTorch is plumbing only:
device buffers, streams, the process group.
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

WORKLOAD = {"world": "random", "seed": 42, "aspect_ratio": "3:2", "image_width": 1200, "image_height": 800, "samples_per_pixel": 500, "max_depth": 50}
# Algorithmic bytes per sample for this workload (DESIGN.md "Roofline"): counts measured by the instrumented oracle on
# the reference BVH topology (tests/test_oracle_golden_images.py keeps them in range), device fp32 record sizes:
# 85.4 box tests x 32 B + 10.8 sphere tests x 20 B + 2.56 rays x 16 B material + 12 B/pixel output / spp.
ALGO_BYTES_PER_SAMPLE = 85.4 * 32 + 10.8 * 20 + 2.56 * 16 + 12.0 / 500
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(pkg, cam19, seconds_budget=20.0):
    """The oracle (kind "port": a C++ restatement - the Rust reference cannot be built here) with the reference's own
    per-row PCG64 streams on a bounded sample of the same workload: evenly spaced rows, all host cores."""
    import numpy as np
    import oracle_binding
    oracle = oracle_binding.Oracle()
    ow = oracle.world(WORKLOAD["world"], WORKLOAD["seed"])
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 32))
    w, h, spp = WORKLOAD["image_width"], WORKLOAD["image_height"], WORKLOAD["samples_per_pixel"]
    rows = np.unique(np.linspace(0, h - 1, min(h, 6 * threads)).astype(np.int32))  # 0.6 Msamples per row: ~13 s on 32 threads
    _, _, stats, secs = ow.render(cam19, w, h, spp, seed=WORKLOAD["seed"], rng_mode=oracle_binding.RNG_PCG_ROW, rows=rows, threads=threads)
    ow.close()
    samples = len(rows) * w * spp
    return {"value": samples / secs / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%d evenly spaced rows of the %dx%d spp=%d image (%.1f Msamples, %.1f s), PCG64 row streams, f64 C++ restatement of the Rust reference"
                      % (len(rows), w, h, spp, samples / 1e6, secs),
            "stats_per_sample": {k: round(v / stats["samples"], 3) for k, v in stats.items() if k != "samples"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (rows gathered through host memory) only to rehearse N>1 on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
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

    W, H, spp = WORKLOAD["image_width"], WORKLOAD["image_height"], WORKLOAD["samples_per_pixel"]
    world = pkg.World(WORKLOAD["world"], WORKLOAD["seed"])
    cam = pkg.default_camera(world, 1.5)
    params = pkg.make_params(world, W, H, spp, WORKLOAD["max_depth"])
    precision = pkg.PRECISION_FP32 if args.precision == "fp32" else pkg.PRECISION_FP64
    renderer = pkg.Renderer(world, cam, params, precision=precision, device=device_index)

    my_rows = rowtile.rows_for_rank(H, rank, world_size)
    d_rows = torch.tensor(my_rows, dtype=torch.int32, device=dev)
    d_rgb = torch.zeros((len(my_rows), W, 3), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    gather_device = torch.device("cpu") if (args.backend == "gloo" and world_size > 1) else dev  # gloo: rehearsal only
    gatherer = rowtile.RowGatherer(H, W, rank, world_size, gather_device)
    frame = {}

    def step():
        renderer.render_device(d_rows.data_ptr(), len(my_rows), d_rgb.data_ptr(), None, stream.cuda_stream)
        # the one collective of the path: finished rows to rank 0 over xGMI (a plain scatter into the frame at N = 1)
        frame["image"] = gatherer.gather(d_rgb if gather_device == dev else d_rgb.cpu())

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dominant kernel: the library's own HIP events around the last launch on the launch stream
    kernel_ms = renderer.last_kernel_ms()
    step_ms_events = [a.elapsed_time(b) for a, b in ev]

    if rank == 0:
        total_samples = W * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        my_samples = len(my_rows) * W * spp
        achieved = ALGO_BYTES_PER_SAMPLE * my_samples / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec (pixels*spp) --world=random 1200x800 spp=500",
            "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
            "config": {"workload": "--world=random --seed=42 --aspect_ratio=3:2 --image_width=1200 --samples_per_pixel=500 (BASELINE.json configs[1])",
                       "rows_per_rank": len(my_rows), "row_tiling": "round-robin rows, one gather", "rng": "philox4x32-10 per (pixel, sample)",
                       "kernel_ms_last_launch": kernel_ms, "step_ms_hip_events": step_ms_events, **renderer.info()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                         "hbm_physical_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "note": "algorithmic bytes of the REFERENCE traversal (reference BVH topology, SURVEY.md 8d) per sample x samples/s: "
                                 "the scene is LDS resident and the device BVH is SAH-built, so this is not physical HBM utilisation and can exceed 1"},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            cam19 = list(cam.origin) + list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) + list(cam.u) + list(cam.v) + [cam.lens_radius]
            out["cpu_baseline"] = cpu_baseline(pkg, cam19)
        print(json.dumps(out), flush=True)
    renderer.close()
    world.close()
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
