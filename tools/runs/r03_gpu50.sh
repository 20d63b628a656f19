#!/bin/bash
# round 3, call 50: compact batches go to the fused kernel as they are (no expansion launch) - parity, then e2e A/B
python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py -m gpu -q -x -k "host_solver or serving or compact or heuristics or dropin or pipeline or solve_many or agent or harness" 2>&1 | tail -2
for d in 1 0 1 0; do
DGCN_HOST_COMPACT_DIRECT=$d python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('direct=$d C3:', round(d['value']), d['ms_per_step'], 'e2e', round(d['e2e']['value']), d['e2e'].get('results_equal_resident_step'))"
done
DGCN_HOST_COMPACT_DIRECT=1 python bench.py --config C4-share --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('direct=1 C4-l20:', round(d['value']), 'e2e', round(d['e2e']['value']), d['e2e'].get('results_equal_resident_step'))"
DGCN_HOST_COMPACT_DIRECT=0 python bench.py --config C4-share --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('direct=0 C4-l20:', round(d['value']), 'e2e', round(d['e2e']['value']), d['e2e'].get('results_equal_resident_step'))"
