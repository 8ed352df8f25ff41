"""Print registers / scratch / occupancy of every render-kernel variant of one kernel build (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [f32|f64|f64s] [extra hipcc flags]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mu-lambda-raytracer_amd", "csrc", "mlrt_kernel.hip")
BUILDS = ("f32", "f64", "f64c", "f64s")
which = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in BUILDS else "f32"
rest = [a for a in sys.argv[1:] if a not in BUILDS]
if which == "f32":
    kflags = ["-DMLRT_REAL=float", "-DMLRT_CULL=float", "-DMLRT_SUFFIX=f32", "-DMLRT_BLOCK=" + os.environ.get("MLRT_BLOCK", "1024"), "-DMLRT_FULL_WAVES_PER_EU=" + os.environ.get("MLRT_FULL_WAVES", "4")] + os.environ.get("MLRT_KERNEL_FLAGS", "").split()
elif which in ("f64", "f64c"):
    kflags = ["-DMLRT_REAL=double", "-DMLRT_CULL=" + ("float" if which == "f64" else "double"), "-DMLRT_SUFFIX=" + which, "-DMLRT_BLOCK=" + os.environ.get("MLRT_BLOCK64", "512"), "-DMLRT_LEAN_BLOCK=" + os.environ.get("MLRT_LEAN_BLOCK64", "1024"),
              "-DMLRT_LEAN_WAVES_PER_EU=" + os.environ.get("MLRT_LEAN_WAVES64", "4"), "-DMLRT_FULL_WAVES_PER_EU=" + os.environ.get("MLRT_FULL_WAVES64", "2")] + (["-DMLRT_LEAN_MATERIALS_HBM=1"] if which == "f64" else []) + os.environ.get("MLRT_KERNEL_FLAGS64", "").split()
else:
    kflags = ["-DMLRT_REAL=double", "-DMLRT_CULL=double", "-DMLRT_STRICT=1", "-DMLRT_SUFFIX=f64s", "-DMLRT_BLOCK=256", "-DMLRT_LEAN_BLOCK=256", "-DMLRT_LEAN_WAVES_PER_EU=1", "-DMLRT_FULL_WAVES_PER_EU=1"]
flags = ["-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "--offload-arch=gfx950"] + kflags + ["-ffp-contract=off",
         "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres_%s.o" % which] + rest
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags, capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name"):
        name = t.split(": ")[1]
        mm = re.search(r"mlrt_render_kernelILb(\d)ELj(\d+)ELi(\d+)", name)
        cur = ("render<lds=%s,feat=%s,block=%s>" % mm.groups()) if mm else name[:40]
        print("\n" + cur, end=": ")
    elif any(t.startswith(k) for k in ("VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "SGPRs:", "VGPR Spill", "SGPR Spill")):
        print(t, end="; ")
print()
