#!/bin/bash
# round 5, call 30: phase clocks of rollout steps of the fused residual kernel on the round-5 build (blocked ranking, unrolled completions)
bash tools/build_diag.sh 2>&1 | grep -i error
for before in 0 20 40 60; do
  DGCN_LIB=distgcn_amd/libdgcn_diag.so STAMP_NOBIAS=1 python tools/stamp_residual.py $before 64 500 rollout 2>/dev/null
done > gpurun_out/r05_residual_step_phases.txt
rm -f distgcn_amd/libdgcn_diag.so
cat gpurun_out/r05_residual_step_phases.txt
