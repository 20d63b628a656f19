#!/bin/bash
# round 6, call 27: C4 share (500 BA test2-mix graphs, 20 layers): the second half of the dispatch order smallest first (option fused_fold) in the
# two-workgroups-per-CU launch (k_fused<.., GW>, forced with fused_gw=1), against the 1 024-thread launch the share takes today
DGCN_AB_KIND=ba python tools/ab_fused.py "fused_gw=0" "fused_gw=1" "fused_gw=1,fused_fold=256" "fused_gw=1,fused_fold=250" "fused_gw=1,fused_fold=244" "fused_gw=1,fused_order=0" 2>&1 | tail -6 | tee gpurun_out/r06_fold.txt
DGCN_AB_KIND=ba DGCN_AB_GRAPHS=400 python tools/ab_fused.py "fused_gw=0" "fused_gw=1" "fused_gw=1,fused_fold=256" "fused_gw=1,fused_fold=200" 2>&1 | tail -4 | tee -a gpurun_out/r06_fold.txt
