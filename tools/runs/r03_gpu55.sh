#!/bin/bash
# round 3, call 55: the compact form as a kernel of its own (by name too) - GPU suite, then the default bench line under the profiler
python -m pytest tests -m gpu -q 2>&1 | tail -2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r03
mkdir -p "$O"; rm -rf "$O/bench_default"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_default" -- python3 $R/bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cd "$R"
f=$(ls $O/bench_default/*/*kernel_stats.csv | head -1); head -5 $f | cut -c1-170
tail -1 $O/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default:', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'e2e', round(d['e2e']['value']), d['e2e']['runs_graphs_per_s'])"
