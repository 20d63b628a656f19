#!/bin/bash
# round 5, call 41: HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of the any-size rollout searches'
# step kernels (k_big<RESID> at l = 20, k_wide1 at l = 1), for bench.py --config MC900-rollout's roofline.traffic
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r05
mkdir -p "$O"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900roll_$c" -- python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 --only rollout > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900rolll1_$c" -- python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 1 --host 0 --only rollout > /dev/null 2>&1
done
ls "$O" | grep roll
