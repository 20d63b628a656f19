#!/bin/bash
# round 5, call 28: the fused residual kernel's ranking (a lane keeps four vertices, a wave meets each w once; cit: a reduction): parity + C5 times
python -m pytest tests/test_gpu_tail.py tests/test_gpu_api.py tests/test_gpu_general.py -m gpu -x -q 2>&1 | tail -4
python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 2>&1 | grep -v '^{"path'
python tools/run_iterative.py --graphs 64 --n 500 --p 0.1 --layers 20 --host 0 2>&1 | grep -v '^{"path'
python bench.py --config C5 2>&1 | tail -1 | cut -c1-600
