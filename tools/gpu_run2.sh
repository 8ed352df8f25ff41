set -x
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
for v in "" _b256 _b768 _b1024; do MLRT_LIB=$L/libmlrt_hip$v.so python tests/gpu_perf.py c2 c4; done > gpurun_out/perf2.log 2>&1
python tests/gpu_diffmap.py final_scene 400 1.0 64 >> gpurun_out/perf2.log 2>&1
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p1 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1
cat gpurun_out/perf2.log
