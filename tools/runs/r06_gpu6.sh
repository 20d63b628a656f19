#!/bin/bash
# round 6, call 6: the 1 024-thread k_fused with eight-row units for long rows (option fused_hub): C4 share, a lone BA(300) graph, ER at 1 024 threads; bits
DGCN_AB_KIND=ba python tools/ab_fused.py "fused_hub=0" "fused_hub=1" 2>&1 | tail -2 | tee gpurun_out/r06_gpu6_ab.txt
DGCN_AB_KIND=ba DGCN_AB_GRAPHS=256 python tools/ab_fused.py "fused_hub=0" "fused_hub=1" 2>&1 | tail -2 | tee -a gpurun_out/r06_gpu6_ab.txt
DGCN_AB_GRAPHS=200 python tools/ab_fused.py "fused_hub=0" "fused_hub=1" 2>&1 | tail -2 | tee -a gpurun_out/r06_gpu6_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06_gpu6_tests.txt
