#!/bin/bash
# round 4, call 31: tail kernel with one barrier per layer / weights a layer ahead: tests, per-step cost, searches
timeout 1500 python -m pytest tests/test_gpu_tail.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu31.log 2>&1
tail -8 gpurun_out/r04_gpu31.log
python tools/tail_probe.py 2>&1 | grep layers
python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{"
