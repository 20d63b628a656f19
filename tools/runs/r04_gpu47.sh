#!/bin/bash
# round 4, call 47: phase clocks of the C3 launch (two workgroups per CU) with the stamps kept in registers
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er200 20 500 2>&1 | grep -v amdgpu > gpurun_out/r04_gpu47.log
cat gpurun_out/r04_gpu47.log
