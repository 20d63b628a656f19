#!/bin/bash
# round 4, call 43: phase clocks of a residual step with the stamps kept in registers (no global round trip per phase)
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for w in rollout cit; do
  for b in 0 40 70; do
    python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu | tail -17
  done
done > gpurun_out/r04_gpu43.log 2>&1
cat gpurun_out/r04_gpu43.log
