#!/bin/bash
# round 4, call 27: the floor of a launch whose graphs have a CU each and almost no vertices (diag build, ablation switches)
DGCN_LIB=distgcn_amd/libdgcn_diag.so DGCN_FUSED_CLUSTER=0 python tools/solo_floor.py 64 > gpurun_out/r04_gpu27.log 2>&1
cat gpurun_out/r04_gpu27.log
