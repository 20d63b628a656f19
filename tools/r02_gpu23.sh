#!/bin/bash
for nl in 2 20; do
echo "== lone, l=$nl"; DGCN_FUSED_CLUSTER=0 DGCN_LIB=$PWD/distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er $nl 1 2>/dev/null | sed -n 2,14p
echo "== cluster 4, l=$nl"; DGCN_FUSED_CLUSTER=4 DGCN_LIB=$PWD/distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er $nl 1 2>/dev/null | sed -n 2,14p
done
