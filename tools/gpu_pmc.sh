# usage: bash tools/gpu_pmc.sh <outdir-tag>   (PMC passes on one C2 render; each pass is its own run)
export TMPDIR=/tmp
T=$1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$T/$n -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$T.$n.log 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_$T/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mlrt" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg): print(f"{k:28s} {agg[k]:.4g}")
if agg.get("SQ_ACTIVE_INST_VALU"): print("VALU lane utilisation", agg["SQ_THREAD_CYCLES_VALU"]/(agg["SQ_ACTIVE_INST_VALU"]*64))
print("wave-VALU-instr per sample", agg["SQ_INSTS_VALU"]/480e6, "SALU per sample", agg["SQ_INSTS_SALU"]/480e6)
PY
