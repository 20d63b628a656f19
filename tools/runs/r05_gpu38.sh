#!/bin/bash
# round 5, call 38: C5 searches with the residual kernel on 512 threads per graph instead of 1 024 (DGCN_FUSED_BLOCK), and per-step times late in a search
for blk in 1024 512; do
  echo "DGCN_FUSED_BLOCK=$blk"
  DGCN_FUSED_BLOCK=$blk python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-330
done
