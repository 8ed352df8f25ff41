# two rocprofv3 --pmc passes (instruction counts, wave-cycle split) on ONE launch of a bench config: bash tools/gpu_pmc_quick.sh <config> <precision> <tag>
export TMPDIR=/tmp
CFG=${1:-c2}; PREC=${2:-fp64}; TAG=${3:-quick}
DT=f64; [ "$PREC" = fp32 ] && DT=f32
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/${TAG}_pmcraw_${CFG}_${DT}
rm -rf $O; mkdir -p $O
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
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
