#!/bin/bash
# round 5, call 17: kernel stats of a rollout search on 64 joint 3 x 300 graphs (l = 20) with the new k_res_cand
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_roll -- python3 $R/tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 --only rollout > /dev/null 2>&1
f=$(find $R/gpurun_out/r05_roll -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-150
