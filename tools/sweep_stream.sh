#!/bin/bash
# A/B: streaming level-0 kernel with different wavefront counts per launch, and the tile kernel
set -e
for w in 1600 2048 2600 3300 4096; do
  GKLT_STREAM_WAVES=$w timeout -k 10 120 python bench.py --steps 1024 --warmup 64 --no-cpu --no-pcie > gpurun_out/sw_$w.json 2>/dev/null
done
timeout -k 10 120 python bench.py --steps 1024 --warmup 64 --no-cpu --no-pcie --pyramid-impl 2 > gpurun_out/sw_tile.json 2>/dev/null
