#!/bin/bash
# round 3, call 45: residual-graph P0 with every row's bounds and first 16 columns requested up front
python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py -m gpu -q -x -k "iterative or rollout or residual or c5 or executed or cgs or dit or cit or cluster or wireless or masked" 2>&1 | tail -2
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220
python tools/run_iterative.py --graphs 8 --host 0 2>&1 | tail -3 | cut -c1-220
python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5:', d['value'], d['ms_per_step'])"
for b in 0 50; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $b 64 500 rollout 2>&1 | grep -v amdgpu.ids | grep "step after\|P0c\|sum of"; done
