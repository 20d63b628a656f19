#!/bin/bash
# round 5, call 33: the rollout step's completions and pick inside the step's launch (rollout_bits.h; k_wide1, k_big, k_big2): parity + step times, off / on
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_general.py tests/test_gpu_big2.py -m gpu -x -q -k "rollout or residual or iterative or beam or nan or wireless or suite" 2>&1 | tail -4
for bits in 0 1; do
  echo "DGCN_ROLLOUT_BITS=$bits"
  for args in "--family mc --graphs 64 --n 900 --p 0.03 --layers 1" "--family mc --graphs 256 --n 900 --p 0.03 --layers 1" "--family mc --graphs 64 --n 900 --p 0.03 --layers 20" "--family mc --graphs 64 --n 1500 --p 0.03 --layers 20" "--graphs 64 --n 500 --p 0.1 --layers 20"; do
    DGCN_ROLLOUT_BITS=$bits timeout 300 python tools/run_iterative.py $args --host 0 --only rollout 2>&1 | grep -v '^{"path\|amdgpu.ids'
  done
done
