#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02f
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 200 python tools/ab_fused.py "" > "$O/ab.log" 2>&1
cat "$O/summary.txt"; tail -6 "$O/pytest.log"; grep median "$O/ab.log"
