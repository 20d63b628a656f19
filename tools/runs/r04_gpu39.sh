#!/bin/bash
# round 4, call 39: k_lgs on bit masks for graphs of 385 .. 1 024 vertices: tests, then iterative solvers with it on / off
timeout 1800 python -m pytest tests/test_gpu_general.py tests/test_gpu_kernels.py tests/test_gpu_tail.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -x -q -p no:cacheprovider 2>&1 | tail -3
for m in 1 0; do
  echo "DGCN_LGS_MASKS=$m"
  DGCN_LGS_MASKS=$m python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" | cut -c1-150
  DGCN_LGS_MASKS=$m python tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" | cut -c1-150
done
