#!/bin/bash
# round 6, call 28: option fused_fold on mixed batches whose images share CUs as they are (ER(n, 0.1), n = 80 .. 200): the ordinary 512-thread launch
DGCN_AB_KIND=ermix python tools/ab_fused.py "" "fused_fold=256" "fused_order=0" 2>&1 | tail -3 | tee gpurun_out/r06_fold_ermix.txt
DGCN_AB_KIND=ermix DGCN_AB_GRAPHS=400 python tools/ab_fused.py "" "fused_fold=256" "fused_order=0" 2>&1 | tail -3 | tee -a gpurun_out/r06_fold_ermix.txt
DGCN_AB_KIND=ermix DGCN_AB_GRAPHS=320 python tools/ab_fused.py "" "fused_fold=256" "fused_order=0" 2>&1 | tail -3 | tee -a gpurun_out/r06_fold_ermix.txt
