#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cluster" 2>&1 | tail -3
python tools/time_small.py 1
for k in 4 6 8; do echo "K=$k"; DGCN_FUSED_CLUSTER=$k python tools/time_small.py 1; done
timeout 300 python tools/cluster_check.py 2>&1 | tail -12
