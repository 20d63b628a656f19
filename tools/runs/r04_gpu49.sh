#!/bin/bash
# round 4, call 49: phase clocks of k_big on the two bench shapes
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for c in er500 mc900; do python tools/stamp_big.py $c 20 256 2>&1 | grep -v amdgpu; done > gpurun_out/r04_gpu49.log
cat gpurun_out/r04_gpu49.log
