#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02m
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "c4_full or test_loop or executed_reference" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
DGCN_LIB=$R/distgcn_amd/libdgcn_diag.so timeout 300 python tools/stamp_fused.py er 20 > "$O/fused_phase_clocks.txt" 2>&1
cat "$O/summary.txt"; tail -4 "$O/pytest.log"; head -22 "$O/fused_phase_clocks.txt"
