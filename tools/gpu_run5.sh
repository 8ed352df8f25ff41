L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
echo v3; python tests/gpu_diffmap.py final_scene 400 1.0 64 2>&1 | grep -v amdgpu; cp gpurun_out/diffmap_final_scene.npy gpurun_out/diffmap_fs_v3.npy
echo v2; MLRT_LIB=$L/libmlrt_hip_v2.so python tests/gpu_diffmap.py final_scene 400 1.0 64 2>&1 | grep -v amdgpu; cp gpurun_out/diffmap_final_scene.npy gpurun_out/diffmap_fs_v2.npy
