#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py --steps 20 --warmup 5 > "$O/bench_20.json" 2> "$O/bench_20.err"; echo "bench20 rc=$?" >> "$O/summary.txt"
timeout 300 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench default rc=$?" >> "$O/summary.txt"
cat "$O/summary.txt"; tail -8 "$O/pytest.log"
python - <<'P'
import json
for f in ("bench_20","bench_default"):
    d=json.loads([l for l in open("gpurun_out/r02d/%s.json"%f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
    print(" e2e", json.dumps(d["e2e"])[:400])
    print(" margin", json.dumps(d["margin_risk"]))
    s=d["spmm_kernel_roofline"]; print(" spmm", s["avg_launch_us"], s["frac"], s["frac_plain_B_spmm"], json.dumps(s.get("out_of_cache"))[:300])
P
