#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
python tools/ab_fused.py "DGCN_FUSED_SPLIT=1" "DGCN_FUSED_SPLIT=0"
DGCN_LIB=$PWD/distgcn_amd/libdgcn_old.so python tools/time_fused.py old
python tools/time_fused.py new
DGCN_LIB=$PWD/distgcn_amd/libdgcn_old.so python tools/time_fused.py old
python tools/time_fused.py new
