#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "supports2 or chebyshev or every_shipped or rejects_wide or predict_state or bench_contract" > "$O/pytest_new.log" 2>&1; echo "pytest new rc=$?" >> "$O/summary.txt"
timeout 600 python tools/tune_spmm_hbm.py > "$O/tune_spmm_hbm.log" 2>&1; echo "tune rc=$?" >> "$O/summary.txt"
cat "$O/summary.txt"; tail -15 "$O/pytest_new.log"; cat "$O/tune_spmm_hbm.log"
