#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests -m gpu -x -q -k "serving or dqn_agent_solve or sharded_solve_single or evaluation_harness or bench_contract" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python tools/ab_fused.py "" "DGCN_FUSED_LANEMAP=1" > "$O/ab_fused.log" 2>&1; echo "ab rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py --steps 20 --warmup 5 > "$O/bench_20.json" 2> "$O/bench_20.err"; echo "bench20 rc=$?" >> "$O/summary.txt"
timeout 400 python tools/tune_spmm_hbm.py "DGCN_SPMM_ROWS=200,DGCN_SPMM_BLOCK=512,DGCN_SPMM_CSRCAP=0" "DGCN_SPMM_PAD=0" "DGCN_SPMM_PAD=8" "DGCN_SPMM_ROWS=200,DGCN_SPMM_BLOCK=1024,DGCN_SPMM_PAD=0" > "$O/tune2.log" 2>&1
cat "$O/summary.txt"; tail -5 "$O/pytest.log"; cat "$O/ab_fused.log"; grep -v "ROWS': '64\|ROWS': '100" "$O/tune2.log"
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r02c/bench_20.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], json.dumps(d["e2e"]))
P
