#!/bin/bash
# round 6, call 9: after the fix of k_big2's sentinel slot (found by the hub / many-tile fuzz arm at case 30 of 250): the five fuzz
# tests at 1 000 cases, then the whole GPU suite
DGCN_FUZZ_CASES=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_fuzz_1000.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_gpu_suite.txt
