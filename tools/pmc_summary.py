"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc.sh for one (config, dtype) into profiles-ready files.

    python tools/pmc_summary.py <dir with <pass>/p_counter_collection.csv> <config> <dtype> <samples per launch> <out prefix>

Counters are summed over the dispatches of the RENDER kernel only (mlrt_render_kernel; the candidate pre-pass and the
resolve kernel are listed separately), one launch per pass (bench.py --steps 1 --warmup 0).  HBM bytes: FETCH_SIZE and
WRITE_SIZE of ALL mlrt kernels of the launch, in their own passes, FETCH_SIZE doubled (gfx950 reports half the bytes of a
wide coalesced read, MI355X_MICROARCH.md "HBM"); both are in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 1024


def main():
    d, config, dtype, samples, prefix = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
    render, other, names = collections.defaultdict(float), collections.defaultdict(float), set()
    launches = collections.defaultdict(set)  # counter -> dispatch ids of the render kernel it was read for
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mlrt" not in k:
                continue
            names.add(k.split("(")[0][:80])
            is_render = "mlrt_render_kernel" in k or "mlrt_split_kernel" in k
            (render if is_render else other)[r["Counter_Name"]] += float(r["Counter_Value"])
            if is_render:
                launches[r["Counter_Name"]].add(r.get("Dispatch_Id", len(launches[r["Counter_Name"]])))
    # a pass may hold several launches of the same frame (bench.py's end_to_end record renders it again): every figure below is PER LAUNCH
    n_launch = max(1, min(len(v) for v in launches.values())) if launches else 1
    if launches and len({len(v) for v in launches.values()}) != 1:
        print("warning: passes hold different numbers of render launches:", {k: len(v) for k, v in launches.items()})
    for table in (render, other):
        for k in table:
            table[k] /= n_launch
    g = lambda n: render.get(n, 0.0)
    cycles = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from build_stamp import stamp
    out = {"config": config, "dtype": dtype, "samples_per_launch": samples, "render_launches_per_pass": n_launch, "build_stamp": stamp(), "kernels": sorted(names),
           "counters_render_kernel": dict(sorted(render.items())), "counters_prepass_and_resolve": dict(sorted(other.items()))}
    if g("SQ_INSTS_VALU"):
        out["valu_insts_per_sample"] = g("SQ_INSTS_VALU") / samples
        out["salu_insts_per_sample"] = g("SQ_INSTS_SALU") / samples
        out["lds_insts_per_sample"] = g("SQ_INSTS_LDS") / samples
    if cycles and g("SQ_INSTS_VALU"):
        out["gpu_cycles"] = cycles
        out["kernel_ms_at_2.4GHz"] = cycles / 2.4e6
        out["valu_issue_frac"] = g("SQ_INSTS_VALU") * 2.0 / (SIMDS * cycles)  # 1 wave-instruction / 2 cycles / SIMD-32
    if g("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
        # lane-level instructions (an instruction counts once per ACTIVE lane): what fewer instructions, not emptier waves, brings down
        out["valu_lane_insts_per_sample"] = out.get("valu_insts_per_sample", 0.0) * 64.0 * out["lane_utilisation"]
        out["valu_active_quads_per_inst"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
    if g("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_WAVE_CYCLES"):
        out["wave_cycle_split"] = {"waiting_s_waitcnt": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "issue_stalled": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                                   "executing": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")}
    mix = {k[len("SQ_INSTS_VALU_"):]: v / samples for k, v in render.items() if k.startswith("SQ_INSTS_VALU_")}
    if mix:
        out["valu_mix_per_sample"] = dict(sorted(mix.items()))
    fetch = render.get("FETCH_SIZE", 0.0) + other.get("FETCH_SIZE", 0.0)
    write = render.get("WRITE_SIZE", 0.0) + other.get("WRITE_SIZE", 0.0)
    if fetch or write:
        out["fetch_size_kib"] = fetch
        out["write_size_kib"] = write
        out["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".txt", "w") as f:
        f.write("# rocprofv3 --pmc passes, per launch of %s (%s) (%d launches of the same frame per pass, averaged), render kernel only unless noted; tools/gpu_pmc.sh + tools/pmc_summary.py\n" % (config, dtype, n_launch))
        f.write("# build: lib sha256 %s, sources sha256 %s\n" % (out["build_stamp"]["lib_sha256"], out["build_stamp"]["source_sha256"]))
        for k, v in sorted(render.items()):
            f.write("%-32s %.6g\n" % (k, v))
        for k in ("gpu_cycles", "valu_issue_frac", "lane_utilisation", "lds_bank_conflict_share", "valu_insts_per_sample", "salu_insts_per_sample", "lds_insts_per_sample",
                  "valu_active_quads_per_inst", "valu_lane_insts_per_sample", "hbm_bytes_per_launch"):
            if k in out:
                f.write("%-32s %.6g\n" % (k, out[k]))
        if "wave_cycle_split" in out:
            f.write("wave cycles: %s\n" % json.dumps(out["wave_cycle_split"]))
        if mix:
            f.write("VALU mix per sample: %s\n" % json.dumps(out["valu_mix_per_sample"]))
    print(open(prefix + ".txt").read())


if __name__ == "__main__":
    main()
