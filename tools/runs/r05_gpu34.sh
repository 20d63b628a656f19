#!/bin/bash
# round 5, call 34: the fused residual kernel's completions all at once, an instance per bit: parity + C5 times
timeout 1500 python -m pytest tests/test_gpu_tail.py tests/test_gpu_api.py tests/test_gpu_general.py tests/test_gpu_wide.py -m gpu -x -q -k "rollout or residual or iterative or beam or search" 2>&1 | tail -4
python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 --only rollout 2>&1 | grep -v '^{"path\|amdgpu.ids'
python tools/run_iterative.py --graphs 256 --n 500 --p 0.02 --layers 20 --host 0 --only rollout 2>&1 | grep -v '^{"path\|amdgpu.ids'
python bench.py --config C5 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-400
