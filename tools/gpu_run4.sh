export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/gputests4.log 2>&1; tail -8 gpurun_out/gputests4.log
for v in "$@"; do MLRT_LIB=$L/libmlrt_hip$v.so timeout -k 10 120 python tests/gpu_perf.py c2 c4; done 2>&1 | grep -v amdgpu.ids
