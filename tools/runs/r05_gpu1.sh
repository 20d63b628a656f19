#!/bin/bash
# round 5, call 1: k_big's row-order sort with one bin per entry count (advisor finding): rows of 575+ entries vs the twin, the any-size suite, timings
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_tail.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r05_gpu1.log 2>&1
tail -5 gpurun_out/r05_gpu1.log
for c in er500 mc900; do python tools/run_general.py $c 300 20 256 2>/dev/null | grep big_solve; done
