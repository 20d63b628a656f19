#!/bin/bash
# round 4, call 8: kernel trace of the any-size path with k_big: ER(500, 0.1) x 256, N = 900 x 256
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_prof3
rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/er500" -- python3 $R/bench.py --graphs 256 --nodes 500 --p 0.1 --layers 20 --steps 100 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool > "$O/er500.json" 2> "$O/er500.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/n900" -- python3 $R/bench.py --graphs 256 --nodes 900 --p 0.01 --layers 20 --steps 100 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool > "$O/n900.json" 2> "$O/n900.err"
cd $R
for d in er500 n900; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 "$f" | cut -c1-60,160-260; done
