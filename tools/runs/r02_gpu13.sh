#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r02
mkdir -p "$O"; rm -rf "$O/bench_layered" "$O/bench_layered.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_layered" -- python3 $R/bench.py --mode layered --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e > "$O/bench_layered.json" 2> "$O/bench_layered.err"
tail -c 600 "$O/bench_layered.json"
