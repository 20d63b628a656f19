#!/bin/bash
# round 5, call 35: what the per-wave weight fetch costs a residual step: phase clocks with the fetch switched off (diag build, DGCN_FUSED_DIAG bit 3: constants instead of weights)
bash tools/build_diag.sh 2>&1 | grep -i error
for d in 0 8; do for before in 20 60; do
  echo "DGCN_FUSED_DIAG=$d"
  DGCN_FUSED_DIAG=$d DGCN_LIB=distgcn_amd/libdgcn_diag.so STAMP_NOBIAS=1 python tools/stamp_residual.py $before 64 500 cit 2>/dev/null | grep "step after\|hidden\|barrier\|wall"
done; done
rm -f distgcn_amd/libdgcn_diag.so
