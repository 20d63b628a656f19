#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
timeout 300 python tools/cluster_check.py 2>&1 | grep -v "N=500\|N= 77"
python tools/run_single.py 400 2>&1 | tail -4
