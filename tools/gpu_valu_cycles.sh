# The VALU microbenchmark under rocprofv3: GRBM_GUI_ACTIVE (cycles the chip was busy, summed over 8 XCDs) and SQ_INSTS_VALU per launch, and
# the kernel durations, so that cycles per instruction do not rest on an assumed clock.  Output: gpurun_out/valu_cycles.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/valu_pmc; rm -rf $O; mkdir -p $O
(cd $GRAFT_REPO_ROOT && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc -o p -- tools/microbench/bin/valu_issue 4 > $O/pmc.log 2>&1)
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- tools/microbench/bin/valu_issue 4 > $O/trace.log 2>&1)
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/valu_cycles.txt
import csv, glob, collections
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
print("# valu_issue at 4 waves per SIMD, 1024 SIMDs, 65536 x 16 instructions per wave: per launch (median over the 8 launches)")
print("# kernel  GRBM_GUI_ACTIVE/8 [cycles]  SQ_INSTS_VALU  cycles per wave-instruction per SIMD  kernel ms (trace)  implied clock GHz")
import statistics as st
for k in sorted(pmc):
    cyc = st.median(pmc[k]["GRBM_GUI_ACTIVE"]) / 8.0
    insts = st.median(pmc[k]["SQ_INSTS_VALU"])
    ms = st.median(dur[k]) if dur.get(k) else float("nan")
    print("%-18s %12.0f %14.0f %8.3f %10.3f %8.3f" % (k[:18], cyc, insts, cyc / (insts / 1024.0), ms, cyc / (ms * 1e6) if ms == ms else float("nan")))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/valu_cycles.txt
