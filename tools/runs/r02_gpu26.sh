#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k cluster 2>&1 | tail -2
timeout 300 python tools/cluster_check.py 7 2>&1 | grep -v "N=500\|N= 77"
python tools/lat_probe.py 2>&1 | tail -4
