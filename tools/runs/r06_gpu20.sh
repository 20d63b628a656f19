#!/bin/bash
# round 6, call 20: k_fused<.., GW> (values and gather words in global scratch: two 512-thread workgroups per CU on the BA mix) against the 1 024-thread launch
DGCN_AB_KIND=ba python tools/ab_fused.py "fused_gw=0" "fused_gw=1" 2>&1 | tail -2 | tee gpurun_out/r06_gw_ab.txt
DGCN_AB_KIND=ba DGCN_AB_GRAPHS=4000 python tools/ab_fused.py "fused_gw=0" "fused_gw=1" 2>&1 | tail -2 | tee -a gpurun_out/r06_gw_ab.txt
python tools/ab_fused.py "fused_gw=0" "fused_gw=1" 2>&1 | tail -2 | tee -a gpurun_out/r06_gw_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06_gw_tests.txt
