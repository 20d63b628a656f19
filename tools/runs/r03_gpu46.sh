#!/bin/bash
# round 3, call 46: C5 evidence once more on the final kernels (kernel-trace stats, residual-step phase clocks, step times)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r03
mkdir -p "$O"; rm -rf "$O/bench_c5" "$O/bench_default"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_c5" -- python3 $R/bench.py --config C5 --cpu-seconds 25 > "$O/bench_c5.json" 2> "$O/bench_c5.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_default" -- python3 $R/bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cd "$R"
bash tools/build_diag.sh > /dev/null 2>&1
for w in rollout cit; do for b in 0 50; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu.ids; done; done > "$O/residual_phase_clocks.txt"
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220 > "$O/iterative_steps.txt"
cat "$O/iterative_steps.txt"
