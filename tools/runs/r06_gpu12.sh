#!/bin/bash
# round 6, call 12: code placement of k_fused - the same source built with different block-alignment flags (distgcn_amd/libdgcn_v*.so,
# built in the container: v1 as the product, v2 -amdgpu-disable-loop-alignment, v3 / v6 / v4 -align-all-nofallthru-blocks=5 / 6 / 7,
# v5 -amdgpu-early-inline-all=false, v7 -align-all-blocks=4), C3 and the C4 share, two rounds each
for r in 1 2; do for v in 1 2 3 6 4 5 7; do echo -n "v$v: "; DGCN_LIB=distgcn_amd/libdgcn_v$v.so python tools/run_fused.py er 300 20 500 2>/dev/null | tail -1; done; done | tee gpurun_out/r06_placement.txt
for v in 1 2 3 6 4 7; do echo -n "v$v ba: "; DGCN_LIB=distgcn_amd/libdgcn_v$v.so python tools/run_fused.py ba 300 20 500 2>/dev/null | tail -1; done | tee -a gpurun_out/r06_placement.txt
