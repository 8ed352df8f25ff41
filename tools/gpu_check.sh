# after a kernel change: hang check (hard timeouts), the GPU suite, then kernel times of the bench workloads
bash tools/gpu_hangcheck.sh 2>&1 | grep -v amdgpu | tail -2 && timeout -k 10 300 python -m pytest tests -m gpu -x -q --timeout 90 > gpurun_out/gputests_check.log 2>&1; tail -4 gpurun_out/gputests_check.log
python tools/gpu_perf.py c2 c3 c4 2>&1 | grep -v amdgpu | cut -c1-150
python tools/gpu_scale_probe.py 1 8 2>&1 | grep -v amdgpu
