#!/bin/bash
# round 3, call 31: persistent double-buffered SpMM (LDS-DMA for the Z slice) - parity, then A/B on the two HBM working sets
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "spmm" 2>&1 | tail -3
timeout 900 python tools/tune_spmm_hbm.py DGCN_SPMM_PERSIST=0 DGCN_SPMM_PERSIST=1 2>&1 | grep -v amdgpu.ids | tail -12
