export TMPDIR=/tmp
T=$1
for pass in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_VALU2"; do
  n=$(echo $pass | cut -d' ' -f2)
  rocprofv3 --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mix_$T/$n -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/mix_$T.$n.log 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float)
for f in glob.glob("gpurun_out/mix_$T/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mlrt" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg): print(f"{k:28s} {agg[k]:.4g}   per sample {agg[k]/480e6:.2f}")
PY
