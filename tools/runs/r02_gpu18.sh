#!/bin/bash
# ablations of k_fused with the -DDGCN_DIAG build: bit0 no gathers, bit1 no transforms, bit2 no greedy rounds
cd "$GRAFT_REPO_ROOT"
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for d in 0 1 2 3 7; do DGCN_FUSED_DIAG=$d python tools/time_fused.py diag=$d 2>&1 | tail -1; done
python tools/stamp_fused.py er 20 2>&1 | sed -n 2,18p
