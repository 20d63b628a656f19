#!/bin/bash
# round 4, call 53: inside an aggregation call (block-level clocks in registers): late residual steps, and the C3 launch
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
python tools/stamp_residual.py 70 64 500 cit 2>&1 | grep -v amdgpu | tail -5
python tools/stamp_residual.py 40 64 500 cit 2>&1 | grep -v amdgpu | tail -3
python tools/stamp_fused.py er200 20 500 2>&1 | grep "last wave\|per block\|hidden A\|hidden T"
