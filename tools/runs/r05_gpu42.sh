#!/bin/bash
# round 5, call 42: the searches' fuzz test at 4 000 cases (another seed range is not offered by the test: more cases of the same stream)
DGCN_FUZZ_CASES=4000 timeout 3300 python -m pytest tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider -k "searches" > gpurun_out/r05_fuzz_4000.log 2>&1
tail -5 gpurun_out/r05_fuzz_4000.log
