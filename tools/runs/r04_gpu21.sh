#!/bin/bash
# round 4, call 21: fuzz of the any-size path (48 cases), then the whole fuzz file at the default size
mkdir -p gpurun_out
DGCN_FUZZ_CASES=48 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider -x -k any_size > gpurun_out/r04_gpu21.log 2>&1
tail -15 gpurun_out/r04_gpu21.log
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
