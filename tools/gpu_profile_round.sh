# Round profile: bench line, kernel-trace stats of the same command, PMC summaries (tools/gpu_pmc.sh) for the headline config in
# both precisions and for the FULL variant (c4), the VALU issue microbenchmark, the strong-scaling probe.
# usage: bash tools/gpu_profile_round.sh [tag] [part]       outputs under gpurun_out/<tag>_*; part 1 = bench, traces, C2 counters,
# microbenchmark; part 2 = C4 / C3 counters, strong-scaling probe (each part fits one gpurun call); no part = everything
export TMPDIR=/tmp
R=${1:-r03}
PART=${2:-all}
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
# (the counters first, copied into the box's profiles/: the bench lines that follow carry a roofline of THIS build, not a stale one)
if [ "$PART" != 2 ]; then
bash tools/gpu_pmc.sh c2 fp64 $R > gpurun_out/${R}_pmc_c2_f64.log 2>&1
bash tools/gpu_pmc.sh c2 fp32 $R > gpurun_out/${R}_pmc_c2_f32.log 2>&1
cp gpurun_out/${R}_pmc_c2_f64.json gpurun_out/${R}_pmc_c2_f64.txt gpurun_out/${R}_pmc_c2_f32.json gpurun_out/${R}_pmc_c2_f32.txt profiles/
python3 bench.py --steps 100 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; tail -c 600 gpurun_out/${R}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -o $R -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_c4 -o ${R}_c4 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config c4 > gpurun_out/${R}_trace_c4.log 2>&1
bash tools/gpu_valu_cycles.sh > gpurun_out/${R}_valu_cycles.log 2>&1
ls gpurun_out/${R}_trace gpurun_out/${R}_trace_c4
fi
if [ "$PART" != 1 ]; then
bash tools/gpu_pmc.sh c4 fp32 $R > gpurun_out/${R}_pmc_c4_f32.log 2>&1
bash tools/gpu_pmc.sh c4 fp64 $R > gpurun_out/${R}_pmc_c4_f64.log 2>&1
bash tools/gpu_pmc.sh c3 fp64 $R > gpurun_out/${R}_pmc_c3_f64.log 2>&1
bash tools/gpu_pmc.sh c3 fp32 $R > gpurun_out/${R}_pmc_c3_f32.log 2>&1
for c in c3 c4; do for d in f64 f32; do cp gpurun_out/${R}_pmc_${c}_${d}.json gpurun_out/${R}_pmc_${c}_${d}.txt profiles/; done; done
python3 bench.py --steps 20 --warmup 3 --config c3 --no-end-to-end > gpurun_out/${R}_bench_c3.json 2> gpurun_out/${R}_bench_c3.err
python3 bench.py --steps 5 --warmup 1 --config c4 > gpurun_out/${R}_bench_c4.json 2> gpurun_out/${R}_bench_c4.err
python3 tools/gpu_scale_probe.py 1 2 4 8 2>&1 | grep -v amdgpu > gpurun_out/${R}_strong_scaling_probe.txt
MLRT_PREC=1 python3 tools/gpu_scale_probe.py 1 2 4 8 2>&1 | grep -v amdgpu >> gpurun_out/${R}_strong_scaling_probe.txt
fi
