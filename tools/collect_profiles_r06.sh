#!/bin/bash
# Runs on the GPU box (gpurun): round-6 evidence.  Un-profiled bench JSON lines AND rocprofv3 kernel-trace stats for the bench
# command and the any-size configurations (MC900-l1: k_wide1; MC1500: k_big2; MC900 / ER500: k_big), the iterative searches,
# the multi-channel slot loop, round 6's shapes (deep stacks narrower than 32 zero-padded onto k_big / k_big2 against the
# layer-by-layer chain they ran as; [I, L, L.L] models through the solve entry point), HBM-traffic PMC passes and the SQ
# counters of k_fused retaken on this build (separate runs, --pmc with --kernel-trace only).  tools/build_diag.sh runs HERE
# (in the container) beforehand: the diag library travels with the snapshot.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r06
rm -rf "$O"; mkdir -p "$O"
cd /tmp
plain() { local name=$1; shift; python3 "$@" 2>/dev/null | tail -1 > "$O/$name.unprofiled.json"; }
prof() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.json" 2> "$O/$name.err"; }
plain bench_default  $R/bench.py
plain bench_mc900_l1 $R/bench.py --config MC900-l1 --cpu-seconds 12 --no-cpu-pool
plain bench_mc1500   $R/bench.py --config MC1500 --cpu-seconds 12 --no-cpu-pool
plain bench_mc900    $R/bench.py --config MC900 --cpu-seconds 12 --no-cpu-pool
plain bench_er500    $R/bench.py --config ER500 --cpu-seconds 12 --no-cpu-pool
plain bench_c2       $R/bench.py --config C2 --cpu-seconds 6 --no-cpu-pool --no-spmm-probe
plain bench_c4_l1    $R/bench.py --config C4-share --layers 1 --cpu-seconds 6 --no-spmm-probe
plain bench_c4_l20   $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe
plain bench_c5       $R/bench.py --config C5 --cpu-seconds 25
plain bench_c5_256   $R/bench.py --config C5 --graphs 256 --cpu-seconds 0
plain bench_c5_512   $R/bench.py --config C5 --graphs 512 --cpu-seconds 0
plain bench_mc900_rollout    $R/bench.py --config MC900-rollout --cpu-seconds 0 --no-cpu-pool
plain bench_mc900_rollout_l1 $R/bench.py --config MC900-rollout --layers 1 --cpu-seconds 0 --no-cpu-pool
{
  for shape in "mc900 20 20 256 16" "mc900 20 4 256 16" "mc900 20 4 256 4" "mc1500 10 20 64 16" "mc900 20 1 256 1 3" "mc900 20 2 256 1 3" "er200x0.1 20 2 500 1 3"; do
    python3 $R/tools/run_general.py $shape 2>/dev/null | grep "kernels per call" | sed 's/^/as built:           /'
    DGCN_OPTIONS="big=0,big2=0" python3 $R/tools/run_general.py $shape 2>/dev/null | grep "kernels per call" | sed 's/^/big = 0, big2 = 0:  /'
  done
} > "$O/narrow_and_poly.txt"
prof bench_default  $R/bench.py
prof bench_mc900_l1 $R/bench.py --config MC900-l1 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe
prof bench_mc1500   $R/bench.py --config MC1500 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe
prof bench_mc900    $R/bench.py --config MC900 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe
prof bench_er500    $R/bench.py --config ER500 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe
prof iterative_mc900 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0
python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > "$O/iterative_mc900.txt"
python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 1 --host 0 2>/dev/null | grep "^{" > "$O/iterative_mc900_l1.txt"
python3 $R/tools/run_iterative.py --family mc --n 1500 --p 0.03 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > "$O/iterative_mc1500.txt"
python3 $R/tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > "$O/iterative_er500.txt"
python3 $R/tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > "$O/iterative_c5.txt"
python3 $R/tools/run_wireless_mc.py 32 300 3 50 1 > "$O/wireless_mc900_l1.txt" 2>/dev/null
python3 $R/tools/run_wireless_mc.py 32 300 3 50 20 > "$O/wireless_mc900_l20.txt" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900l1_$c" -- python3 $R/tools/run_general.py mc900 5 1 256 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc1500_$c" -- python3 $R/tools/run_general.py mc1500 5 20 256 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_er500_$c" -- python3 $R/tools/run_general.py er500 5 20 256 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900_$c" -- python3 $R/tools/run_general.py mc900 5 20 256 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c3_$c" -- python3 $R/tools/run_fused.py er 5 20 500 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_c5_$c" -- python3 $R/tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 --only rollout > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900roll_$c" -- python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 --only rollout > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_mc900rolll1_$c" -- python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 1 --host 0 --only rollout > /dev/null 2>&1
done
cd "$R"
bash tools/collect_pmc.sh > /dev/null 2>&1
[ -f distgcn_amd/libdgcn_diag.so ] || bash tools/build_diag.sh 2>&1 | grep -i error
DGCN_LIB=distgcn_amd/libdgcn_diag.so python3 tools/stamp_fused.py er 20 500 > "$O/fused_phase_clocks.txt" 2>/dev/null
DGCN_LIB=distgcn_amd/libdgcn_diag.so python3 tools/stamp_big2.py mc1500 20 256 > "$O/big2_phase_clocks.txt" 2>/dev/null
DGCN_LIB=distgcn_amd/libdgcn_diag.so python3 tools/stamp_big2.py er1000x0.01 20 256 >> "$O/big2_phase_clocks.txt" 2>/dev/null
DGCN_LIB=distgcn_amd/libdgcn_diag.so python3 tools/stamp_big.py er500 20 256 > "$O/big_phase_clocks.txt" 2>/dev/null
DGCN_LIB=distgcn_amd/libdgcn_diag.so python3 tools/stamp_big.py mc900 20 256 >> "$O/big_phase_clocks.txt" 2>/dev/null
ls "$O"
