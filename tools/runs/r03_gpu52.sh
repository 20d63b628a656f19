#!/bin/bash
# round 3, call 52: direct compact path - whole GPU suite (cluster forced too: every workgroup of a graph reads the compact batch), e2e lines
python -m pytest tests -m gpu -q 2>&1 | tail -2
DGCN_FUSED_CLUSTER=8 python -m pytest tests/test_gpu_api.py -m gpu -q -k "host_solver or serving or compact or dropin or agent" 2>&1 | tail -1
DGCN_FUSED_ORDER=1 python -m pytest tests/test_gpu_api.py -m gpu -q -k "host_solver or serving or compact or dropin or agent" 2>&1 | tail -1
for c in C3 C2 C4-share; do python bench.py --config $c --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c:', round(d['value']), d['ms_per_step'], 'e2e', round(d['e2e']['value']), d['e2e'].get('results_equal_resident_step'))"; done
python tools/run_single.py 300 2>&1 | tail -4
