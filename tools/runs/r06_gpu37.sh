#!/bin/bash
# round 6, call 37: phase clocks of one C5 rollout step (diag build) early / in the middle of a search, cluster form on (automatic) and off
for before in 5 40; do
  STAMP_NOBIAS=1 DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $before 64 500 rollout 2>&1 | grep -v amdgpu.ids
  STAMP_NOBIAS=1 DGCN_OPTIONS=fused_cluster=0 DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $before 64 500 rollout 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_c5_step_clocks.txt
