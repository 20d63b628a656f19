#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_rollc -- python3 $R/tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 1 --host 0 --only rollout > /dev/null 2>&1
f=$(find $R/gpurun_out/r05_rollc -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-150
cd $R; python tools/run_general.py mc1500 100 20 256 2>/dev/null | grep -v path
