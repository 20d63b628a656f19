#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02j
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_cabi.py -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python tools/run_single.py 500 > "$O/single.log" 2>&1
timeout 300 python tools/run_single.py 300 prof > "$O/single_prof.log" 2>&1
cat "$O/summary.txt"; tail -4 "$O/pytest.log"; cat "$O/single.log"; grep -A30 "cumulative" "$O/single_prof.log" | head -45
