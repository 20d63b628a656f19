#!/bin/bash
# round 6, call 41: the fuzz file at 600 cases per test on the final library
DGCN_FUZZ_CASES=600 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06_fuzz_600.txt
