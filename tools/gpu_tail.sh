# fixed (non-scaling) part of the kernel time vs tile size: 1/1 and 1/8 of the rows
for ts in 512 256 128 64; do echo "== MLRT_TILE_SAMPLES=$ts"; MLRT_TILE_SAMPLES=$ts python tests/gpu_scale_probe.py 1 8 2>&1 | grep -v amdgpu; done | tee gpurun_out/tail_tiles.log
