#!/bin/bash
# round 3, call 14: SpMM launch-geometry sweep on working sets beyond the Infinity Cache
DGCN_TUNE_FULL=1 python tools/tune_spmm_hbm.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_spmm_geometry_sweep.txt
