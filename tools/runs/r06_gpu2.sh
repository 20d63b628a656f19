#!/bin/bash
# round 6, call 2: k_fused with the pulled layer loop (fused_pipe=1) against the lock-step loop, one process, interleaved; bits equal
python tools/ab_fused.py "fused_pipe=0" "fused_pipe=1" 2>&1 | tail -3 | tee gpurun_out/r06_gpu2_ab.txt
DGCN_AB_KIND=ba python tools/ab_fused.py "fused_pipe=0" "fused_pipe=1" 2>&1 | tail -3 | tee -a gpurun_out/r06_gpu2_ab.txt
for v in 1 0; do DGCN_OPTIONS="fused_pipe=$v" DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er 20 500 2>&1 | tail -32 > gpurun_out/r06_gpu2_stamps_pipe$v.txt; done
head -22 gpurun_out/r06_gpu2_stamps_pipe1.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06_gpu2_tests.txt
