#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats of bench.py in both modes + HBM traffic PMC passes.
# PMC passes are separate runs with --kernel-trace only (no sys/hip trace), as the pool requires.
export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/profiles
rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_fused" -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 > "$O/bench_fused.json" 2> "$O/bench_fused.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_layered" -- python3 bench.py --mode layered --steps 20 --warmup 3 --cpu-seconds 0 --no-spmm-probe > "$O/bench_layered.json" 2> "$O/bench_layered.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/iterative" -- python3 tools/run_iterative.py --graphs 64 --host 0 > "$O/iterative.json" 2> "$O/iterative.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_fused_$c" -- python3 tools/run_fused.py er 5 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmm_$c" -- python3 tools/run_spmm.py er 5 > /dev/null 2>&1
done
find "$O" -name "*.csv" | head -40
