#!/bin/bash
# LDS micro-benchmark (incl. metadata-style broadcast reads) and k_fused with entry values in global memory
cd "$GRAFT_REPO_ROOT"
./tools/micro/lds_gather 2>&1 | grep -v amdgpu.ids
python tools/ab_fused.py "" 2>&1 | tail -1
DGCN_FUSED_GVALS=1 python tools/ab_fused.py "" 2>&1 | tail -1
