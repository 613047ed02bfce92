#!/bin/bash
# A/B all in-tree library variants in one process each (same GPU, back to back, two rounds)
mkdir -p gpurun_out
for round in 1 2; do
for lib in nerf-rs_amd/libnerf_mi355x*.so; do
  NERF_DEBUG_CLOCK=1 NERF_MI355X_LIB=$PWD/$lib timeout -k 10 120 python3 tools/quick_bench.py 3 800 ${DTYPE:-f32} 2>&1 | tail -1 | tee -a gpurun_out/ab.log
done; done
