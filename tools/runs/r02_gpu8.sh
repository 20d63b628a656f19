#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02h
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "spmm or layer or forward or every_shipped or full_size or large_graph" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py --mode layered --steps 300 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e > "$O/bench_layered.json" 2> "$O/bench_layered.err"
timeout 300 python tools/tune_spmm_hbm.py "DGCN_SPMM_PAD=4" 2>&1 | grep -v "ROWS" > "$O/tune.log"
cat "$O/summary.txt"; tail -4 "$O/pytest.log"; cat "$O/tune.log"
python - <<'P'
import json
for f in ("bench_layered",):
    d=json.loads([l for l in open("gpurun_out/r02h/%s.json"%f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], {k:(round(v["avg_us"],1), v["launches_per_step"]) for k,v in d["kernels"].items()}, d["roofline"])
P
