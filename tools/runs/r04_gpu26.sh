#!/bin/bash
# round 4, call 26: where a step of a nearly finished search spends its time (diag build: phase clocks of the residual kernel)
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for w in rollout cit; do
  for b in 0 60 100; do
    python tools/stamp_residual.py $b 64 500 $w 2>&1 | tail -16
  done
done > gpurun_out/r04_gpu26.log 2>&1
tail -50 gpurun_out/r04_gpu26.log
