#!/bin/bash
# round 6, call 8: the hub / many-tile fuzz arm at 250 cases, with the failing case's tag
DGCN_FUZZ_CASES=1000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider -k hub_rows 2>&1 | grep -E "assert|tag|Error|error|^E " | head -30 | tee gpurun_out/r06_fuzz_hub_fail.txt
