#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02k
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q -k "executed_reference" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
cat "$O/summary.txt"; tail -30 "$O/pytest.log"
