#!/bin/bash
# round 6, call 4: the new tests first ([I, L, L.L] in the solve entry points, restatement-forward searches on k_big2<RESID> and the
# two-layer k_wide1, the hub / many-tile fuzz arm), then the whole GPU suite with DGCN_FUZZ_CASES at its new default of 100
timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_big2.py tests/test_gpu_wide.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider \
  -k "three_support or restatement_forward or hub_rows or narrow" 2>&1 | tail -15 | tee gpurun_out/r06_gpu4_new_tests.txt
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=12 2>&1 | tail -25 | tee gpurun_out/r06_gpu4_suite.txt
