#!/bin/bash
# round 3, call 56: C4 (l = 20) and C5 bench lines under the profiler with the final kernel names
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r03
mkdir -p "$O"; rm -rf "$O/bench_c4_l20" "$O/bench_c5" "$O/bench_c4_full"
cd /tmp
prof() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"; }
prof bench_c4_l20  $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe
prof bench_c5      $R/bench.py --config C5 --cpu-seconds 25
prof bench_c4_full $R/bench.py --config C4 --layers 20 --steps 150 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 40
cd "$R"; ls $O | grep -c json
