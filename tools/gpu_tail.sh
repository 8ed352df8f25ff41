# fixed (non-scaling) part of the kernel time vs the tapering passes
for tt in ${TAILS:-none 64,64,128 64,64,128,128}; do [ $tt = none ] && tt=""; echo "== MLRT_TILE_TAIL=$tt"; MLRT_TILE_TAIL=$tt python tools/gpu_scale_probe.py ${RANKS:-1 8} 2>&1 | grep -v amdgpu; done | tee gpurun_out/tail_tiles.log
