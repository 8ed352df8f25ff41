"""What a profile was measured on: sha256 of the built library and of the sources it is built from.

    from build_stamp import stamp     ->  {"lib_sha256": ..., "source_sha256": ...}
    python tools/build_stamp.py profiles/r03_*.json   (in the git checkout: adds "git_head" to summaries whose lib_sha256 is the local library's)

tools/pmc_summary.py writes the stamp into every profiles/*.json on the GPU box (which has no .git: the commit is added
here, when the summary is copied into profiles/); bench.py compares it with the library it is about to time and reports
roofline.profile_stale when the SOURCES differ - the counters per sample in the committed summary then describe another kernel -
and roofline.profile_same_binary when even the binary is the one that was profiled (two hipcc runs on one source do not give the
same bytes, so a rebuild changes the library's hash and nothing else).
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mu-lambda-raytracer_amd")
SOURCES = sorted(glob.glob(os.path.join(PKG, "csrc", "*")) + [os.path.join(PKG, "build.py"), os.path.join(ROOT, "include", "mlrt.h")])


def _sha(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stamp(lib_path=None):
    lib_path = lib_path or os.environ.get("MLRT_LIB") or os.path.join(PKG, "lib", "libmlrt_hip.so")
    return {"lib_sha256": _sha([lib_path]) if os.path.exists(lib_path) else None, "source_sha256": _sha(SOURCES)}


if __name__ == "__main__":
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "mu-lambda-raytracer_amd", "include"], capture_output=True, text=True).stdout.strip()
    here = stamp()
    for path in sys.argv[1:]:
        doc = json.load(open(path))
        st = doc.get("build_stamp") or {}
        if st.get("source_sha256") == here["source_sha256"]:  # (same sources; hipcc's output is not bit-reproducible, so the binary may be a rebuild)
            st["same_binary_as_checkout"] = st.get("lib_sha256") == here["lib_sha256"]
            st["git_head"] = head + ("+uncommitted" if dirty else "")
            doc["build_stamp"] = st
            json.dump(doc, open(path, "w"), indent=1)
            print(path, "->", st["git_head"])
        else:
            print(path, ": measured on another build (lib %s..., sources %s...); not stamped" % (str(st.get("lib_sha256"))[:12], str(st.get("source_sha256"))[:12]))
