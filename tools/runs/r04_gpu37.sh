#!/bin/bash
# round 4, call 37: the fuzz tests with many cases (tail on / off on both paths; plain solves down the any-size path)
DGCN_FUZZ_CASES=150 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu37.log 2>&1
tail -15 gpurun_out/r04_gpu37.log
