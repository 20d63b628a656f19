#!/bin/bash
# round 6, call 21: where k_fused<.., GW> starts to pay on the BA mix (graphs per launch)
for n in 768 1000 1500 2000; do echo "BA x $n:"; DGCN_AB_KIND=ba DGCN_AB_GRAPHS=$n python tools/ab_fused.py "fused_gw=0" "fused_gw=1" 2>&1 | tail -2; done | tee gpurun_out/r06_gw_sizes.txt
