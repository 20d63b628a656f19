#!/bin/bash
# round 6, call 36: workgroup-by-workgroup timeline of the C4 share's launch (diag build)
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/timeline_fused.py ba 20 500 2>&1 | tee gpurun_out/r06_c4_timeline.txt
DGCN_LIB=distgcn_amd/libdgcn_diag.so DGCN_OPTIONS=fused_gw=1 python tools/timeline_fused.py ba 20 500 2>&1 | tee -a gpurun_out/r06_c4_timeline.txt
