timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/gputests6.log 2>&1; tail -4 gpurun_out/gputests6.log
MLRT_STATS=1 python tests/gpu_perf.py c2 c4 2>&1 | grep -v amdgpu | cut -c1-330
echo "--- no SAH"; MLRT_STATS=1 MLRT_NO_SAH=1 python tests/gpu_perf.py c2 c4 2>&1 | grep -v amdgpu | cut -c1-330
