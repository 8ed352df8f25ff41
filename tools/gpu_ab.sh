# interleaved A/B of all lib variants: 3 rounds
L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
for round in 1 2 3; do for f in $L/libmlrt_hip*.so; do MLRT_LIB=$f python tools/gpu_perf.py ${WHICH:-c2}; done; done 2>&1 | grep -v amdgpu | cut -c1-130 | sort | awk '{print}'
