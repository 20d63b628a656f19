#!/bin/bash
# round 6, call 35 (experiment; needed a build with option fused_order_vw, not kept - the change is the key's vertex weight in k_graph_rank / fused_order_bits):
DGCN_AB_KIND=ba python tools/ab_fused.py "" "fused_order_vw=4" "fused_order_vw=32" "fused_order_vw=48" "fused_order_vw=72" "fused_order_vw=128" "fused_order_vw=512" 2>&1 | tail -7 | tee gpurun_out/r06_order_vw.txt
DGCN_AB_KIND=ba DGCN_AB_GRAPHS=4000 python tools/ab_fused.py "" "fused_order_vw=4" "fused_order_vw=32" "fused_order_vw=72" "fused_order_vw=128" 2>&1 | tail -5 | tee -a gpurun_out/r06_order_vw.txt
DGCN_AB_KIND=ba DGCN_AB_GRAPHS=1000 python tools/ab_fused.py "" "fused_order_vw=4" "fused_order_vw=32" "fused_order_vw=72" "fused_order_vw=128" 2>&1 | tail -5 | tee -a gpurun_out/r06_order_vw.txt
