#!/bin/bash
# round 4, call 1: the any-size device path (csrc/general.hip) - first run of its tests
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu1.log 2>&1
tail -5 gpurun_out/r04_gpu1.log
