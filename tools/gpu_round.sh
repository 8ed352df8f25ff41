# GPU suite + round profile + strong-scaling probe, for the kernels as committed
python -m pytest tests -m gpu -x -q > gpurun_out/gputests_$1.log 2>&1; tail -3 gpurun_out/gputests_$1.log
bash tools/gpu_profile_round.sh $1 && python tools/gpu_scale_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/scale_$1.log
