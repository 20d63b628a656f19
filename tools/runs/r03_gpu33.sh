#!/bin/bash
# round 3, call 33: is the weight-fragment fetch what the transform of a small residual graph waits for?  (diag build, fetch switched off)
for d in 0 8; do echo "== DGCN_FUSED_DIAG=$d"; DGCN_FUSED_DIAG=$d DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py 50 64 500 cit 2>&1 | grep -v amdgpu.ids | grep "step after\|hidden\|barrier\|sum of"; done
for d in 0 8; do echo "== DGCN_FUSED_DIAG=$d"; DGCN_FUSED_DIAG=$d DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py 0 64 500 cit 2>&1 | grep -v amdgpu.ids | grep "step after\|hidden\|barrier\|sum of"; done
