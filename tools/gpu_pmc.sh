# usage: bash tools/gpu_pmc.sh <config c2|c3|c4|c5> <precision fp64|fp32> [tag]
# rocprofv3 --pmc passes (each its own run, counters only) on ONE launch of a bench workload, summarised into
# gpurun_out/<tag>_pmc_<config>_<dtype>.{json,txt} (copy the ones to be judged into profiles/).
export TMPDIR=/tmp
CFG=${1:-c2}; PREC=${2:-fp64}; TAG=${3:-r03}
DT=f64; [ "$PREC" = fp32 ] && DT=f32
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/${TAG}_pmcraw_${CFG}_${DT}
rm -rf $O; mkdir -p $O
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_ATOMIC" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $ROOT && rocprofv3 --pmc $pass --output-format csv -d $O/pass$i -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast --no-end-to-end --config $CFG --precision $PREC > $O/pass$i.log 2>&1) || { echo "pass $i failed"; tail -5 $O/pass$i.log; }
done
SAMPLES=$(python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
c = bench.CONFIGS["$CFG"]; w = c["image_width"]; h = int(w / c["aspect"])
print(w * h * c["samples_per_pixel"])
PY
)
python3 $ROOT/tools/pmc_summary.py $O $CFG $DT $SAMPLES $ROOT/gpurun_out/${TAG}_pmc_${CFG}_${DT}
