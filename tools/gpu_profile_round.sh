# Round profile: bench line, kernel-trace stats of the same command, PMC summaries (tools/gpu_pmc.sh) for the headline config in
# both precisions and for the FULL variant (c4), the VALU issue microbenchmark, the strong-scaling probe.
# usage: bash tools/gpu_profile_round.sh [tag]       outputs under gpurun_out/<tag>_*
export TMPDIR=/tmp
R=${1:-r03}
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
python3 bench.py --steps 100 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; tail -c 600 gpurun_out/${R}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -o $R -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_c4 -o ${R}_c4 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config c4 > gpurun_out/${R}_trace_c4.log 2>&1
bash tools/gpu_pmc.sh c2 fp64 $R > gpurun_out/${R}_pmc_c2_f64.log 2>&1
bash tools/gpu_pmc.sh c2 fp32 $R > gpurun_out/${R}_pmc_c2_f32.log 2>&1
bash tools/gpu_pmc.sh c4 fp32 $R > gpurun_out/${R}_pmc_c4_f32.log 2>&1
bash tools/gpu_pmc.sh c4 fp64 $R > gpurun_out/${R}_pmc_c4_f64.log 2>&1
bash tools/gpu_pmc.sh c3 fp64 $R > gpurun_out/${R}_pmc_c3_f64.log 2>&1
bash tools/gpu_pmc.sh c3 fp32 $R > gpurun_out/${R}_pmc_c3_f32.log 2>&1
bash tools/gpu_valu_cycles.sh > gpurun_out/${R}_valu_cycles.log 2>&1
python3 tools/gpu_scale_probe.py 1 2 4 8 2>&1 | grep -v amdgpu > gpurun_out/${R}_strong_scaling_probe.txt
MLRT_PREC=1 python3 tools/gpu_scale_probe.py 1 2 4 8 2>&1 | grep -v amdgpu >> gpurun_out/${R}_strong_scaling_probe.txt
ls gpurun_out/${R}_trace gpurun_out/${R}_trace_c4
