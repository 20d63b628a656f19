#!/bin/bash
# round 3, call 53: default bench line three times (e2e: median of three runs each)
for i in 1 2 3; do python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', round(d['value']), d['ms_per_step'], 'e2e', round(d['e2e']['value']), d['e2e']['runs_graphs_per_s'], d['e2e'].get('results_equal_resident_step'))"; done
