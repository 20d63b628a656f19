#!/bin/bash
# round 3, call 54: the bench lines of the final build under the profiler once more (default, C2, C4 share at both depths)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r03
mkdir -p "$O"; rm -rf "$O/bench_default" "$O/bench_c2" "$O/bench_c4_l1" "$O/bench_c4_l20"
cd /tmp
prof() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"; }
prof bench_default $R/bench.py
prof bench_c2      $R/bench.py --config C2 --cpu-seconds 6 --no-cpu-pool --no-spmm-probe
prof bench_c4_l1   $R/bench.py --config C4-share --layers 1 --cpu-seconds 6 --no-spmm-probe
prof bench_c4_l20  $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe
cd "$R"; ls "$O" | head -20
