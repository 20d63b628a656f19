#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02l
rm -rf "$O"; mkdir -p "$O"
cd "$R"
( echo "# __graft_entry__.build(force=True) on the GPU box ($(hostname), $(date -u +%FT%TZ))"; rocminfo 2>/dev/null | grep -m2 "gfx950\|Marketing" ; timeout 900 python -c "import __graft_entry__ as g; g.build(force=True); print('build ok'); g.smoke()" ) > "$O/build_on_gpu_box.log" 2>&1; echo "build rc=$?" >> "$O/summary.txt"
cat distgcn_amd/BUILD_INFO.json >> "$O/build_on_gpu_box.log"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
timeout 400 python bench.py --steps 20 --warmup 5 > "$O/bench_20.json" 2> "$O/bench_20.err"; echo "bench rc=$?" >> "$O/summary.txt"
cat "$O/summary.txt"; tail -4 "$O/pytest.log"; tail -12 "$O/build_on_gpu_box.log"
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r02l/bench_20.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["e2e"]["value"], d["margin_risk"])
s=d["spmm_kernel_roofline"]; print(s["frac"], s["frac_plain_B_spmm"], s["out_of_cache"]["frac"], s["out_of_cache_one_launch"])
P
