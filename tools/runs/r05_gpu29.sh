#!/bin/bash
# round 5, call 29: rollout completions with four neighbour chains in flight: parity + C5 times
python -m pytest tests/test_gpu_tail.py tests/test_gpu_api.py -m gpu -x -q -k "rollout or residual or iterative" 2>&1 | tail -3
python tools/run_iterative.py --graphs 64 --n 500 --p 0.02 --layers 20 --host 0 --only rollout 2>&1 | grep -v '^{"path'
python tools/run_iterative.py --graphs 256 --n 500 --p 0.02 --layers 20 --host 0 --only rollout 2>&1 | grep -v '^{"path'
