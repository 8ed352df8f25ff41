# in-kernel counters of the debug build (shader clocks per loop section, lane utilisation) for the bench workloads
MLRT_STATS=1 MLRT_LIB=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib/libmlrt_hip_debug.so python tools/gpu_perf.py ${WHICH:-c2 c4} 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/stats.log
