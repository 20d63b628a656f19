#!/bin/bash
# round 3, call 57: last gate - forced variants on the host-to-host tests, fuzz, smoke
DGCN_FUSED_CLUSTER=8 python -m pytest tests/test_gpu_api.py -m gpu -q -k "host_solver or serving or compact or dropin or agent" 2>&1 | tail -1
DGCN_FUSED_ORDER=1 python -m pytest tests/test_gpu_api.py -m gpu -q -k "host_solver or serving or compact or dropin or agent" 2>&1 | tail -1
DGCN_HOST_COMPACT_DIRECT=0 python -m pytest tests/test_gpu_api.py -m gpu -q -k "host_solver or serving or compact or dropin or agent" 2>&1 | tail -1
DGCN_FUZZ_CASES=100 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
