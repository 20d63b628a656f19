#!/bin/bash
# round 4, call 32: tail kernel per-step cost by phase (diag build)
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/tail_probe.py > gpurun_out/r04_gpu32.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_gpu32.log
