L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
for f in $L/libmlrt_hip*.so; do MLRT_LIB=$f timeout -k 10 120 python tests/gpu_perf.py ${WHICH:-c2 c4}; done 2>&1 | grep -v amdgpu.ids | cut -c1-150
