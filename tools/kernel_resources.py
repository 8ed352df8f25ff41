"""Print registers / scratch / occupancy of every render-kernel variant of the f32 build (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mu-lambda-raytracer_amd", "csrc", "mlrt_kernel.hip")
flags = ["-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-DMLRT_REAL=float", "-DMLRT_SUFFIX=f32",
         "-DMLRT_BLOCK=" + os.environ.get("MLRT_BLOCK", "1024"), "-DMLRT_FULL_WAVES_PER_EU=" + os.environ.get("MLRT_FULL_WAVES", "4"), "-ffp-contract=off",
         "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + os.environ.get("MLRT_KERNEL_FLAGS", "").split() + sys.argv[1:]
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
