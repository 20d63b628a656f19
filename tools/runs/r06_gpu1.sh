#!/bin/bash
# round 6, call 1: the whole GPU suite after the option-table refactoring (no getenv left in the library), baseline C3 timing
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r06_gpu1_suite.txt
python tools/run_fused.py er 300 20 500 2>/dev/null | tail -1 > gpurun_out/r06_gpu1_c3.txt
python tools/run_fused.py ba 300 20 500 2>/dev/null | tail -1 >> gpurun_out/r06_gpu1_c3.txt
cat gpurun_out/r06_gpu1_suite.txt gpurun_out/r06_gpu1_c3.txt
