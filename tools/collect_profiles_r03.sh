#!/bin/bash
# Runs on the GPU box (gpurun): round-3 evidence.  Kernel-trace stats of the bench command and of every BASELINE config
# (bench.py --config ...); HBM-traffic PMC passes per config (separate runs, --pmc with --kernel-trace only, as the pool requires).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r03
rm -rf "$O"; mkdir -p "$O"
cd /tmp
prof() { # name, then the python command line
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"
}
prof bench_default $R/bench.py
prof bench_c2      $R/bench.py --config C2 --cpu-seconds 6 --no-cpu-pool --no-spmm-probe
prof bench_c4_l1   $R/bench.py --config C4-share --layers 1 --cpu-seconds 6 --no-spmm-probe
prof bench_c4_l20  $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe
prof bench_c4_full $R/bench.py --config C4 --layers 20 --steps 150 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 40
prof bench_c5      $R/bench.py --config C5 --cpu-seconds 25
prof bench_layered $R/bench.py --mode layered --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e
python3 $R/bench.py --two-streams --cpu-seconds 0 --no-spmm-probe --parity-seconds 0 > "$O/bench_two_streams.json" 2> /dev/null  # (no trace: side figure only)
prof spmm_cache    $R/tools/run_spmm.py er 30 1
prof spmm_rot8     $R/tools/run_spmm.py er 6 8
prof spmm_one4000  $R/tools/run_spmm.py er 6 -8
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c3_$c" -- python3 $R/tools/run_fused.py er 5 20 500 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c2_$c" -- python3 $R/tools/run_fused.py er100 5 1 500 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c4l1_$c" -- python3 $R/tools/run_fused.py ba 5 1 500 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c4l20_$c" -- python3 $R/tools/run_fused.py ba 5 20 500 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmm_$c" -- python3 $R/tools/run_spmm.py er 5 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmmrot8_$c" -- python3 $R/tools/run_spmm.py er 3 8 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_spmmone4000_$c" -- python3 $R/tools/run_spmm.py er 3 -8 > /dev/null 2>&1
done
cd "$R"
bash tools/collect_pmc.sh > "$O/collect_pmc.log" 2>&1                       # SQ / LDS counters of k_fused on C3 -> gpurun_out/pmc
bash tools/collect_pmc_cfg.sh c2 er100 1 500 > "$O/collect_pmc_c2.log" 2>&1    # ... of the one-layer kernel on C2 -> gpurun_out/pmc_c2
bash tools/collect_pmc_cfg.sh c4l1 ba 1 500 > "$O/collect_pmc_c4l1.log" 2>&1
bash tools/build_diag.sh > /dev/null 2>&1
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er 20 500 2>&1 | grep -v amdgpu.ids > "$O/fused_phase_clocks.txt"
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids > "$O/shallow_phase_clocks.txt"
find "$O" -name "*kernel_stats.csv" | head -20
du -sh "$O" "$R/gpurun_out/pmc" "$R/gpurun_out/pmc_c2"
