#!/bin/bash
# Runs on the GPU box (gpurun): round-2 evidence.  Kernel-trace stats of the bench command and of every BASELINE config
# quoted in DESIGN.md; HBM-traffic PMC passes (separate runs, --pmc with --kernel-trace only, as the pool requires).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r02
rm -rf "$O"; mkdir -p "$O"
cd /tmp
prof() { # name, then the python command line
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"
}
prof bench_fused   $R/bench.py
prof bench_layered $R/bench.py --mode layered --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e
prof bench_c2      $R/bench.py --nodes 100 --layers 1 --steps 500 --warmup 5 --cpu-seconds 3 --no-cpu-pool --no-spmm-probe
prof bench_c4_l1   $R/bench.py --family ba --layers 1 --steps 500 --warmup 5 --cpu-seconds 0 --no-spmm-probe
prof bench_c4_l20  $R/bench.py --family ba --layers 20 --steps 300 --warmup 5 --cpu-seconds 0 --no-spmm-probe
prof c5_iterative  $R/tools/run_iterative.py --graphs 64 --host 0
python3 $R/bench.py --two-streams --cpu-seconds 0 --no-spmm-probe > "$O/bench_two_streams.json" 2> /dev/null  # (no trace: side figure only)
prof spmm_cache    $R/tools/run_spmm.py er 30 1
prof spmm_rot8     $R/tools/run_spmm.py er 6 8
prof spmm_one4000  $R/tools/run_spmm.py er 6 -8
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_fused_$c" -- python3 $R/tools/run_fused.py er 5 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmm_$c" -- python3 $R/tools/run_spmm.py er 5 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmmrot8_$c" -- python3 $R/tools/run_spmm.py er 3 8 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmmone4000_$c" -- python3 $R/tools/run_spmm.py er 3 -8 > /dev/null 2>&1
done
cd "$R"
bash tools/collect_pmc.sh > "$O/collect_pmc.log" 2>&1
find "$O" -name "*kernel_stats.csv" | head -20
du -sh "$O" "$R/gpurun_out/pmc"
