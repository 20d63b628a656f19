#!/bin/bash
# round 5, call 2: wide.hip (one-layer models on graphs of any size, one launch) - its tests, then the any-size suite
timeout 1500 python -m pytest tests/test_gpu_wide.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r05_gpu2.log 2>&1
tail -25 gpurun_out/r05_gpu2.log
timeout 900 python -m pytest tests/test_gpu_general.py -x -q --tb=short -p no:cacheprovider 2>&1 | tail -5
