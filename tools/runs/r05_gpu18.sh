#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for d in 1 2 0; do
cd /tmp && DGCN_CAND_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_roll$d -- python3 $R/tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 --only rollout > /dev/null 2>&1
f=$(find $R/gpurun_out/r05_roll$d -name "*kernel_stats.csv" | head -1); echo "dbg $d: $(grep k_res_cand $f | cut -c1-120)"
done
