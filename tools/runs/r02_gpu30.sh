#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cluster" 2>&1 | tail -2
timeout 300 python tools/cluster_check.py 2>&1 | grep "layers 20" | grep "N=500\|N=300\|B=  1"
timeout 300 python tools/cluster_check.py 8 2>&1 | grep "layers 20" | grep "N=500\|N=300"
