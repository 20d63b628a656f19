#!/bin/bash
# round 3, call 15: the whole GPU suite with each launch variant forced on wherever it is eligible
echo "== DGCN_FUSED_CLUSTER=8"; DGCN_FUSED_CLUSTER=8 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== DGCN_FUSED_ORDER=1"; DGCN_FUSED_ORDER=1 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== DGCN_SHALLOW=0"; DGCN_SHALLOW=0 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== DGCN_HOST_COMPACT=0"; DGCN_HOST_COMPACT=0 python -m pytest tests -m gpu -q -k "host_solver or serving or compact or heuristics or dropin" 2>&1 | tail -3
echo "== fuzz x 200"; DGCN_FUZZ_CASES=200 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
