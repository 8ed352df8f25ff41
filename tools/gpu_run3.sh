export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
python -m pytest tests -x -q -m gpu > gpurun_out/gputests3.log 2>&1; tail -8 gpurun_out/gputests3.log
for v in "" _b768 _b1024; do MLRT_LIB=$L/libmlrt_hip$v.so python tests/gpu_perf.py c2 c4; done 2>&1 | grep -v amdgpu.ids
python tests/gpu_diffmap.py final_scene 400 1.0 64 2>&1 | grep -v amdgpu.ids
