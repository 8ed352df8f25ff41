"""Registers / scratch / occupancy of every render-kernel variant of the four kernel builds, with the flags of build.py, compiled into
/tmp (the shipped library is not touched): python tools/kernel_resources_all.py > profiles/rNN_kernel_resources.txt"""
import importlib.util, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mlrt_build", os.path.join(ROOT, "mu-lambda-raytracer_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)


def one(unit):
    obj, src, extra = unit
    cmd = [B.HIPCC, "-Rpass-analysis=kernel-resource-usage"] + B.COMMON + extra + ["-c", src, "-o", "/tmp/kres_all_" + obj]
    return obj, subprocess.run(cmd, capture_output=True, text=True).stderr


units = [u for u in B.UNITS if u[1] == B.KERNEL]
with ThreadPoolExecutor(max_workers=4) as pool:
    outs = list(pool.map(one, units))
print("# hipcc -Rpass-analysis=kernel-resource-usage for every render-kernel variant of the four builds, with the flags of mu-lambda-raytracer_amd/build.py (tools/kernel_resources_all.py)")
print("# build lds feat block pair | VGPRs; scratch bytes/lane (f64 non-sphere-only variants: 32 of them are the stack of the out-of-line sphere_uv); waves/SIMD the register allocator was held to")
rows = []
for obj, out in outs:
    build = obj[len("kernel_"):-2]
    cur, rec = None, {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name"):
            name = t.split(": ")[1]
            mm = re.search(r"mlrt_render_kernelILb(\d)ELj(\d+)ELi(\d+)ELb(\d)", name)
            ms = re.search(r"mlrt_split_kernelILb(\d)ELi(\d+)", name)
            cur = ("%-5s lds=%s feat=%3s block=%4s pair=%s" % ((build,) + mm.groups())) if mm else (("%-5s lds=%s split-role kernel block=%4s" % ((build,) + ms.groups())) if ms else None)
            rec = {}
        elif cur:
            for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
                if t.startswith(key + ":"):
                    rec[key] = t.split(": ")[1]
            if len(rec) == 3:
                rows.append("%s | VGPRs: %s; scratch: %s; waves: %s" % (cur, rec["VGPRs"], rec["ScratchSize [bytes/lane]"], rec["Occupancy [waves/SIMD]"]))
                cur = None
for r in sorted(set(rows)):
    print(r)
