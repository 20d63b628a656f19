#!/bin/bash
python -m pytest tests/test_gpu_api.py -x -q -k "iterative or rollout or c5 or wireless or dit or cit or residual" 2>&1 | tail -3
python -m pytest tests/test_gpu_kernels.py -x -q -k "residual or masked or iterative or cluster" 2>&1 | tail -2
for g in 1 8 32; do
  echo "== $g graph(s), cluster off"; DGCN_FUSED_CLUSTER=0 python tools/run_iterative.py --graphs $g --host 0 2>&1 | tail -3 | cut -c1-150
  echo "== $g graph(s), default"; python tools/run_iterative.py --graphs $g --host 0 2>&1 | tail -3 | cut -c1-150
done
