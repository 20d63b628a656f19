#!/bin/bash
# round 6, call 7: the round's evidence (tools/collect_profiles_r06.sh), then the fuzz tests at 1 000 cases each
bash tools/collect_profiles_r06.sh > gpurun_out/r06_collect.log 2>&1
tail -3 gpurun_out/r06_collect.log
DGCN_FUZZ_CASES=1000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06_fuzz_1000.txt
