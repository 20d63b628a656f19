#!/bin/bash
# round 5, call 32: the searches' host loop with the next group of steps on the stream before the last group's progress words are read: parity + times, off / on
python -m pytest tests/test_gpu_tail.py tests/test_gpu_api.py tests/test_gpu_wide.py tests/test_gpu_general.py -m gpu -x -q -k "rollout or residual or iterative or search or dit or cit" 2>&1 | tail -3
for pl in 0 1; do
  echo "DGCN_RESIDUAL_PIPELINE=$pl"
  DGCN_RESIDUAL_PIPELINE=$pl python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-200
  DGCN_RESIDUAL_PIPELINE=$pl python tools/run_iterative.py --family mc --graphs 64 --n 900 --p 0.03 --layers 1 --host 0 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-200
  DGCN_RESIDUAL_PIPELINE=$pl python tools/run_iterative.py --family mc --graphs 64 --n 900 --p 0.03 --layers 20 --host 0 --only rollout 2>&1 | grep -v '^{"path\|amdgpu.ids' | cut -c1-200
done
