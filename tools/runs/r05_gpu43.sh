#!/bin/bash
# round 5, call 43: cand_select with a vertex per lane: places by one v_readlane pass; rollout_bits without global reads: parity + step times
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_general.py tests/test_gpu_big2.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rollout or residual or iterative or beam or nan or wireless or suite or searches" 2>&1 | tail -3
for args in "--family mc --graphs 64 --n 900 --p 0.03 --layers 1" "--family mc --graphs 64 --n 900 --p 0.03 --layers 20" "--family mc --graphs 64 --n 1500 --p 0.03 --layers 20" "--graphs 64 --n 500 --p 0.1 --layers 20"; do
  timeout 300 python tools/run_iterative.py $args --host 0 --only rollout 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-300
done
