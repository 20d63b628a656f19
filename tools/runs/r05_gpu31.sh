#!/bin/bash
# round 5, call 31: the residual kernel's next-layer weights staged once per workgroup by LDS-DMA: parity + C5 times, staging off / on
python -m pytest tests/test_gpu_tail.py tests/test_gpu_api.py -m gpu -x -q -k "rollout or residual or iterative" 2>&1 | tail -3
for st in 0 1; do
  echo "DGCN_FUSED_WSTAGE=$st"
  DGCN_FUSED_WSTAGE=$st python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 2>&1 | grep -v '^{"path\|amdgpu.ids'
done
