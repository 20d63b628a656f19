#!/bin/bash
# round 3, call 35: what the aggregation would cost without LDS bank conflicts (diag build, neighbour parities forced)
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/ablate_fused.py 2>&1 | grep -v amdgpu.ids
