L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
for f in $L/libmlrt_hip.so $L/libmlrt_hip_b512.so $L/libmlrt_hip_b1024.so; do MLRT_LIB=$f python tests/gpu_perf.py c2 c4; done 2>&1 | grep -v amdgpu | cut -c1-150
for o in none small eye; do echo "--- order $o"; MLRT_STATS=1 MLRT_SAH_ORDER=$o MLRT_EYE=13,2,3 python tests/gpu_perf.py c2 2>&1 | grep -v amdgpu | cut -c1-200; done
