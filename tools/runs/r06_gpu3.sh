#!/bin/bash
# round 6, call 3: k_fused layer loops (0 lock-step, 1 ready queue + tickets, 2 owner computes), narrow deep stacks on k_big / k_big2
python tools/ab_fused.py "fused_pipe=0" "fused_pipe=1" "fused_pipe=2" 2>&1 | tail -3 | tee gpurun_out/r06_gpu3_ab.txt
DGCN_AB_KIND=ba python tools/ab_fused.py "fused_pipe=0" "fused_pipe=1" "fused_pipe=2" 2>&1 | tail -3 | tee -a gpurun_out/r06_gpu3_ab.txt
DGCN_OPTIONS="fused_pipe=2" DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er 20 500 2>&1 | tail -32 > gpurun_out/r06_gpu3_stamps_pipe2.txt
head -20 gpurun_out/r06_gpu3_stamps_pipe2.txt
timeout 900 python -m pytest tests/test_gpu_general.py -x -q -p no:cacheprovider -k "narrow" 2>&1 | tail -15 | tee gpurun_out/r06_gpu3_tests.txt
