#!/bin/bash
# round 4, call 52: late residual steps of a model without biases (the bench's C5 model)
export DGCN_LIB=distgcn_amd/libdgcn_diag.so STAMP_NOBIAS=1
for w in cit rollout; do for b in 40 70; do python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu | tail -16; done; done > gpurun_out/r04_gpu52.log
grep "step after\|hidden T\|hidden A\|barrier after\|wall time" gpurun_out/r04_gpu52.log
