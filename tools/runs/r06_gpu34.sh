#!/bin/bash
# round 6, call 34: the fuzz file with the mixed many-graphs arm (largest first / folded order), default depth and 400 cases per test
python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/r06_fuzz_mixed.txt
DGCN_FUZZ_CASES=400 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_shapes_match_the_twin" 2>&1 | tail -2 | tee -a gpurun_out/r06_fuzz_mixed.txt
