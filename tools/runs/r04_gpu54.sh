#!/bin/bash
# round 4, call 54: the fuzz tests with 1 000 cases each (tail on / off on both paths, plain solves down the any-size path, fused path, host solver)
DGCN_FUZZ_CASES=1000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu54.log 2>&1
tail -15 gpurun_out/r04_gpu54.log
