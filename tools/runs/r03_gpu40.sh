#!/bin/bash
# round 3, call 40: iterative solvers with fewer progress read-backs
python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "iterative or rollout or residual or c5 or executed or cgs or dit or cit" 2>&1 | tail -2
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220
python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5:', d['value'], d['ms_per_step'])"
