#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02o
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q -k "wireless or iterative or residual or executed_reference or rollout" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
cat "$O/summary.txt"; tail -25 "$O/pytest.log"
