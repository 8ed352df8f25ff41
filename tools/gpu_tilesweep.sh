# big-tile size sweep (MLRT_TILE_SAMPLES) on the strong-scaling probe
for ts in ${SIZES:-256 384 512 768}; do echo "== MLRT_TILE_SAMPLES=$ts"; MLRT_TILE_SAMPLES=$ts python tools/gpu_scale_probe.py ${RANKS:-1 8} 2>&1 | grep -v amdgpu; done | tee gpurun_out/tilesweep.log
