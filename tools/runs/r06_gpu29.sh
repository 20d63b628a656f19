#!/bin/bash
# round 6, call 29: fused_fold automatic - kernel tests, the mixed batches again (automatic / off), C3 and the C4 share untouched
python -m pytest tests/test_gpu_kernels.py tests/test_cabi.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06_fold_tests.txt
DGCN_AB_KIND=ermix python tools/ab_fused.py "" "fused_fold=0" 2>&1 | tail -2 | tee -a gpurun_out/r06_fold_tests.txt
DGCN_AB_KIND=ermix DGCN_AB_GRAPHS=300 python tools/ab_fused.py "" "fused_fold=0" 2>&1 | tail -2 | tee -a gpurun_out/r06_fold_tests.txt
DGCN_AB_KIND=ba python tools/ab_fused.py "" "fused_fold=0" 2>&1 | tail -2 | tee -a gpurun_out/r06_fold_tests.txt
python tools/ab_fused.py "" "fused_fold=0" 2>&1 | tail -2 | tee -a gpurun_out/r06_fold_tests.txt
