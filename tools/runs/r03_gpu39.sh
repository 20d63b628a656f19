#!/bin/bash
# round 3, call 39: long fuzz on the final build (600 cases) + the whole suite once more
DGCN_FUZZ_CASES=600 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | tail -2
