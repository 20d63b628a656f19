#!/bin/bash
# round 3, call 51: e2e timeline with the direct compact path (kernel trace)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/trace_e2e
rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/t" -- python3 $R/bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 > "$O/bench.json" 2> "$O/bench.err"
cd "$R"
for i in 1 2 3; do python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', round(d['value']), d['ms_per_step'], 'e2e', round(d['e2e']['value']), {k: v for k, v in d['e2e'].items() if k not in ('value',)})" | cut -c1-600; done
