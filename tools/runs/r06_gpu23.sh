#!/bin/bash
# round 6, call 23: C3 with three 320-thread workgroups per CU (the GW image of a 200-vertex graph is 53 KB) - experiment
python tools/ab_fused.py "" "fused_gw=1" "fused_gw=320" 2>&1 | tail -3 | tee gpurun_out/r06_gw320.txt
DGCN_AB_GRAPHS=768 python tools/ab_fused.py "" "fused_gw=1" "fused_gw=320" 2>&1 | tail -3 | tee -a gpurun_out/r06_gw320.txt
DGCN_AB_GRAPHS=1500 python tools/ab_fused.py "" "fused_gw=1" "fused_gw=320" 2>&1 | tail -3 | tee -a gpurun_out/r06_gw320.txt
