#!/bin/bash
# one-graph latency: phase clocks of the lone-workgroup kernel, launch time with and without the cluster variant
bash tools/build_diag.sh
echo "== lone workgroup (DGCN_FUSED_CLUSTER=0), diag build, B=1 er l=20"
DGCN_FUSED_CLUSTER=0 DGCN_LIB=$PWD/distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er 20 1 2>&1 | head -22
echo "== product build launch time B=1"
DGCN_FUSED_CLUSTER=0 python tools/time_small.py 1
python tools/time_small.py 1
for k in 2 4 6 8; do echo "K=$k"; DGCN_FUSED_CLUSTER=$k python tools/time_small.py 1; done
