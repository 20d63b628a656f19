#!/bin/bash
# Runs on the GPU box (gpurun): round-4 evidence.  Un-profiled bench JSON lines AND kernel-trace stats for the bench command
# and the configurations beyond the fused kernel (ER500, MC900); HBM-traffic PMC passes of k_big (separate runs, --pmc with
# --kernel-trace only, as the pool requires).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r04
rm -rf "$O"; mkdir -p "$O"
cd /tmp
plain() { local name=$1; shift; python3 "$@" 2>/dev/null | tail -1 > "$O/$name.unprofiled.json"; }
prof() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"; }
plain bench_default $R/bench.py
plain bench_er500   $R/bench.py --config ER500 --cpu-seconds 15 --no-cpu-pool
plain bench_mc900   $R/bench.py --config MC900 --cpu-seconds 15 --no-cpu-pool
plain bench_c2      $R/bench.py --config C2 --cpu-seconds 6 --no-cpu-pool --no-spmm-probe
plain bench_c4_l1   $R/bench.py --config C4-share --layers 1 --cpu-seconds 6 --no-spmm-probe
plain bench_c4_l20  $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe
plain bench_c5      $R/bench.py --config C5 --cpu-seconds 25
plain bench_c5_256  $R/bench.py --config C5 --graphs 256 --cpu-seconds 0
plain bench_c3_anysize $R/bench.py --any-size-path --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe
plain bench_layered $R/bench.py --mode layered --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e
prof bench_default $R/bench.py
prof bench_er500   $R/bench.py --config ER500 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0
prof bench_mc900   $R/bench.py --config MC900 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0
prof bench_c3_anysize $R/bench.py --any-size-path --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe --parity-seconds 0
prof iterative_mc900 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0
python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 2>/dev/null | grep "^{" > "$O/iterative_mc900.txt"
python3 $R/tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 2>/dev/null | grep "^{" > "$O/iterative_er500.txt"
python3 $R/tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > "$O/iterative_c5.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_er500_$c" -- python3 $R/tools/run_general.py er500 5 20 256 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900_$c" -- python3 $R/tools/run_general.py mc900 5 20 256 > /dev/null 2>&1
done
cd "$R"
ls "$O"
