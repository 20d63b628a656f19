#!/bin/bash
# round 6, call 38: the folded order beyond one round (forced: fused_fold = 256 reverses every position from 256 on) on mixed ER batches of 640 / 768 / 1 024 graphs
for n in 600 768 1024; do
  DGCN_AB_KIND=ermix DGCN_AB_GRAPHS=$n python tools/ab_fused.py "" "fused_fold=256" "fused_fold=512" 2>&1 | tail -3
done | tee gpurun_out/r06_fold_beyond.txt
