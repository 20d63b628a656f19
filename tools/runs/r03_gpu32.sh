#!/bin/bash
# round 3, call 32: where a step of the rollout / cit search spends its time (diag build), start and middle of the search
bash tools/build_diag.sh > /dev/null 2>&1
for w in rollout cit; do for b in 0 50; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu.ids; done; done
