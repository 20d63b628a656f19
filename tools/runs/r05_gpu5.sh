#!/bin/bash
# round 5, call 5: k_big2 (977 .. 1 920 vertices in one launch): its tests, MC1500 timing
timeout 2400 python -m pytest tests/test_gpu_big2.py -x -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/r05_gpu5.log 2>&1
tail -30 gpurun_out/r05_gpu5.log
for c in mc1500; do python tools/run_general.py $c 100 20 256 2>/dev/null | grep -v path; done
