#!/bin/bash
# round 5, call 23: the rollout's completions (k_lgs on beam x graphs masked instances): 1 024-thread against 256-thread workgroups
for blk in 1024 256; do for nl in 1 20; do echo -n "DGCN_LGS_BLOCK=$blk l=$nl: "; DGCN_LGS_BLOCK=$blk python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers $nl --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-330; done; done
