#!/bin/bash
# round 4, call 51: phase clocks of late residual steps after the idle waves stopped fetching weights
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for w in cit rollout; do for b in 40 70; do python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu | tail -16; done; done > gpurun_out/r04_gpu51.log
cat gpurun_out/r04_gpu51.log
