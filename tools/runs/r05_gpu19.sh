#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_wide.py tests/test_gpu_big2.py -x -q --tb=short -p no:cacheprovider -k "rollout or residual or iterative or witness or agree" 2>&1 | tail -4
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_rollb -- python3 $R/tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 --only rollout > /dev/null 2>&1
f=$(find $R/gpurun_out/r05_rollb -name "*kernel_stats.csv" | head -1); head -7 $f | cut -c1-150
cd $R
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-300
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 1 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-300
python tools/run_iterative.py --graphs 8 --n 5000 --p 0.001 --layers 1 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-300
