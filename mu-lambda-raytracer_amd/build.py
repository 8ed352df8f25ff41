"""Build libmlrt_hip.so (HIP kernels for gfx950 + C ABI + host mirror) and the `mlrt` CLI, in-tree.

    python mu-lambda-raytracer_amd/build.py [--force] [--debug]

--debug builds lib/libmlrt_hip_debug.so next to the release library: the same code with the in-kernel counters
(MLRT_STATS=1) and the per-sample path printer (MLRT_DEBUG_PIXEL=pixel,sample) compiled in; load it with
MLRT_LIB=<path>.  The release kernels carry no instrumentation (it costs 8 % even when switched off).

hipcc cross-compiles gfx950 without a GPU.  Outputs: mu-lambda-raytracer_amd/lib/libmlrt_hip.so and
mu-lambda-raytracer_amd/bin/mlrt (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")
BIN_DIR = os.path.join(HERE, "bin")
CLI = os.path.join(BIN_DIR, "mlrt")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# experiment hooks: MLRT_TAG builds a side-by-side variant (lib/libmlrt_hip<tag>.so, loaded with MLRT_LIB=<path>);
# MLRT_KERNEL_FLAGS appends flags to the f32 kernel compile (e.g. "-DMLRT_X=1")
DEBUG = "--debug" in sys.argv
if DEBUG:
    os.environ["MLRT_TAG"] = "_debug"
TAG = os.environ.get("MLRT_TAG", "")
EXTRA_KERNEL_FLAGS = os.environ.get("MLRT_KERNEL_FLAGS", "").split()
EXTRA_KERNEL_FLAGS64 = os.environ.get("MLRT_KERNEL_FLAGS64", "").split()
# the instrumented kernels need the registers the release kernels give up for occupancy: looser launch bounds
DEBUG_FLAGS = ["-DMLRT_DEBUG_BUILD=1", "-DMLRT_PER_VARIANT_SHAPES=0"] if DEBUG else []  # (one loose workgroup shape for every non-sphere-only variant)
DEBUG_FLAGS_F32 = ["-DMLRT_LEAN_BLOCK=1024", "-DMLRT_LEAN_WAVES_PER_EU=4"] if DEBUG else []
if DEBUG:
    os.environ.setdefault("MLRT_BLOCK", "768")
    os.environ.setdefault("MLRT_FULL_WAVES", "3")
BLOCK = os.environ.get("MLRT_BLOCK", "1024")
# workgroup shapes of the f64 product builds: FULL block / LEAN block / LEAN waves per SIMD / FULL waves per SIMD
F64_SHAPE = ["-DMLRT_BLOCK=" + os.environ.get("MLRT_BLOCK64", "512"), "-DMLRT_LEAN_BLOCK=" + os.environ.get("MLRT_LEAN_BLOCK64", "1024"),
             "-DMLRT_LEAN_WAVES_PER_EU=" + os.environ.get("MLRT_LEAN_WAVES64", "4"), "-DMLRT_FULL_WAVES_PER_EU=" + os.environ.get("MLRT_FULL_WAVES64", "2"),
             # the sphere-only variant's materials (a fifth of `random`'s f64 blob) stay in HBM / L2: C2 f64 56.6 -> 55.9 ms (the f32 build: +-0)
             "-DMLRT_LEAN_MATERIALS_HBM=1"]
F64C_SHAPE = ["-DMLRT_BLOCK=512", "-DMLRT_LEAN_BLOCK=1024", "-DMLRT_LEAN_WAVES_PER_EU=4", "-DMLRT_FULL_WAVES_PER_EU=2"]
OBJ_DIR = os.path.join(HERE, "lib", "obj" + TAG)
LIB = os.path.join(LIB_DIR, "libmlrt_hip%s.so" % TAG)

COMMON = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", os.path.join(ROOT, "include")]
KERNEL = os.path.join(HERE, "csrc", "mlrt_kernel.hip")
UNITS = [
    # (object, source, extra flags)
    # Both kernel builds pin FMA contraction OFF: every f32/f64 operation is the IEEE operation the source spells out
    # (the f64 geometry code fuses explicitly, see dfma), so the image does not depend on the compiler's inlining or on
    # which kernel variant runs.  Measured cost against -ffp-contract=fast: 1.5 % (C2) to 3.5 % (final_scene).
    # (MLRT_LEAN_MATERIALS_HBM: with the sphere-only variant's materials read from L2, two 768-thread workgroups - scene copy, accumulators and
    # the pair walk's stacks each - fit the 160 KB of a CU: 6 waves per SIMD; with the materials staged too only one fits: C2 49.8 instead of 37.8 ms)
    ("kernel_f32.o", KERNEL, ["--offload-arch=" + ARCH, "-DMLRT_REAL=float", "-DMLRT_CULL=float", "-DMLRT_SUFFIX=f32", "-DMLRT_BLOCK=" + BLOCK, "-DMLRT_FULL_WAVES_PER_EU=" + os.environ.get("MLRT_FULL_WAVES", "4"), "-DMLRT_LEAN_MATERIALS_HBM=1", "-ffp-contract=" + os.environ.get("MLRT_FP_CONTRACT", "off")] + EXTRA_KERNEL_FLAGS + DEBUG_FLAGS + DEBUG_FLAGS_F32),
    # reference-precision build (product mode): every rounding of a sample's arithmetic where the reference has it; f32
    # culling entries with the primitives' f64 parameters in payload slots; big persistent workgroups, LDS-resident scene
    ("kernel_f64.o", KERNEL, ["--offload-arch=" + ARCH, "-DMLRT_REAL=double", "-DMLRT_CULL=float", "-DMLRT_SUFFIX=f64"] + F64_SHAPE + ["-ffp-contract=off"] + EXTRA_KERNEL_FLAGS64 + DEBUG_FLAGS),
    # the same with f64 culling entries (mlrt_options.cull_f64)
    ("kernel_f64c.o", KERNEL, ["--offload-arch=" + ARCH, "-DMLRT_REAL=double", "-DMLRT_CULL=double", "-DMLRT_SUFFIX=f64c"] + F64C_SHAPE + ["-ffp-contract=off"] + DEBUG_FLAGS),
    # strict build (test instrument): additionally the reference's BVH topology and slab arithmetic
    ("kernel_f64s.o", KERNEL, ["--offload-arch=" + ARCH, "-DMLRT_REAL=double", "-DMLRT_CULL=double", "-DMLRT_STRICT=1", "-DMLRT_SUFFIX=f64s", "-DMLRT_BLOCK=256", "-DMLRT_LEAN_BLOCK=256",
                               "-DMLRT_LEAN_WAVES_PER_EU=1", "-DMLRT_FULL_WAVES_PER_EU=1", "-ffp-contract=off"] + DEBUG_FLAGS),
    ("api.o", os.path.join(HERE, "csrc", "mlrt_api.cpp"), ["-ffp-contract=off"] + DEBUG_FLAGS),
    ("multi.o", os.path.join(HERE, "csrc", "mlrt_multi.cpp"), []),
    ("multi_kernels.o", os.path.join(HERE, "csrc", "mlrt_multi_kernels.hip"), ["--offload-arch=" + ARCH]),
    ("host_api.o", os.path.join(HERE, "host", "host_api.cpp"), ["-ffp-contract=off"]),
    ("worlds.o", os.path.join(HERE, "host", "worlds.cpp"), ["-ffp-contract=off"]),
    ("jpeg.o", os.path.join(HERE, "host", "jpeg_baseline.cpp"), ["-ffp-contract=off"]),
]
HEADERS = [os.path.join(ROOT, "include", "mlrt.h"), os.path.join(ROOT, "include", "mlrt_host.h"), os.path.join(HERE, "csrc", "mlrt_device.hpp"), os.path.join(HERE, "csrc", "mlrt_split.inc"),
           os.path.join(HERE, "host", "host_internal.hpp"), os.path.join(HERE, "host", "pcg64.hpp"), os.path.abspath(__file__)]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose_resources=False):
    for d in (LIB_DIR, BIN_DIR, OBJ_DIR):
        os.makedirs(d, exist_ok=True)
    objs, cmds = [], []
    for obj, src, extra in UNITS:
        out = os.path.join(OBJ_DIR, obj)
        objs.append(out)
        if force or _stale(out, [src] + HEADERS):
            cmd = [HIPCC] + COMMON + extra + ["-c", src, "-o", out]
            if verbose_resources and src == KERNEL:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            cmds.append(cmd)
    if cmds:  # the three kernel builds dominate: compile the stale units side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(cmds), int(os.environ.get("MLRT_BUILD_JOBS", "6")))) as pool:
            list(pool.map(_run, cmds))
    if force or _stale(LIB, objs):
        _run([HIPCC, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])  # (librccl.so is dlopen'ed by the multi-device gather, not linked)
    main_src = os.path.join(HERE, "host", "main.cpp")
    if TAG:
        return LIB
    if force or _stale(CLI, [main_src, LIB] + HEADERS):
        _run([HIPCC] + COMMON + ["-ffp-contract=off", main_src, "-o", CLI, "-L", LIB_DIR, "-lmlrt_hip", "-Wl,-rpath,$ORIGIN/../lib"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose_resources="--resources" in sys.argv)
    print("built", LIB, "and", CLI)
