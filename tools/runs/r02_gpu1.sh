#!/bin/bash
# round-2 GPU pass 1: tests, bench at N=1, the --gpus 2 refusal on a 1-GPU box, RCCL with one rank, SpMM out-of-cache counters
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench default rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 > "$O/bench_gpus2.out" 2> "$O/bench_gpus2.err"; echo "bench --gpus 2 rc=$? (must be non-zero on a 1-GPU box)" >> "$O/summary.txt"
timeout 300 python bench.py --force-dist --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe > "$O/bench_forcedist.json" 2> "$O/bench_forcedist.err"; echo "bench --force-dist rc=$?" >> "$O/summary.txt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bench" -- python3 $R/bench.py --steps 200 --warmup 5 --cpu-seconds 0 > "$O/prof_bench.json" 2> "$O/prof_bench.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmm_ooc_$c" -- python3 $R/tools/run_spmm.py er 3 8 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_spmm_ooc" -- python3 $R/tools/run_spmm.py er 6 8 > /dev/null 2>&1
cat "$O/summary.txt"
tail -3 "$O/pytest.log"
tail -c 1500 "$O/bench_gpus2.err"
