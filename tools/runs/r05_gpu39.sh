#!/bin/bash
# round 5, call 39: C5 searches with the residual kernel's cluster variant forced (K workgroups per graph), now that the step's
# greedy part is short: does spreading the forward over K CUs pay?
for k in 0 2 3 4; do
  echo "DGCN_FUSED_CLUSTER=$k"
  DGCN_FUSED_CLUSTER=$k python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-330
done
