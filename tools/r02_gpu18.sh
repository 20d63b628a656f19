#!/bin/bash
# ablations of k_fused with the -DDGCN_DIAG build: bit0 no gathers, bit1 no transforms, bit2 no greedy rounds,
# bit4 gathers without FMAs, bit5 FMAs without gathers
cd "$GRAFT_REPO_ROOT"
python tools/ab_fused.py "" 2>&1 | tail -1
export DGCN_LIB=distgcn_amd/libdgcn_diag.so
for d in 0 2 3 18 34 50; do DGCN_FUSED_DIAG=$d python tools/time_fused.py diag=$d 2>&1 | tail -1; done
