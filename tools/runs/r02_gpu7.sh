#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02g
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py --mode layered --steps 300 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e > "$O/bench_layered.json" 2> "$O/bench_layered.err"
DGCN_LAYER_FUSE=0 timeout 300 python bench.py --mode layered --steps 300 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e > "$O/bench_layered_unfused.json" 2> "$O/bench_layered_unfused.err"
cat "$O/summary.txt"; tail -6 "$O/pytest.log"
python - <<'P'
import json
for f in ("bench_layered","bench_layered_unfused"):
    d=json.loads([l for l in open("gpurun_out/r02g/%s.json"%f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], {k:(round(v["avg_us"],1), v["launches_per_step"]) for k,v in d["kernels"].items()})
P
