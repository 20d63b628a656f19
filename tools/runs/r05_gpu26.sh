#!/bin/bash
# round 5, call 26 (run again as call 37 on the final kernels): the fuzz tests at 1 000 cases each (plain solves down the fused and the any-size path incl. k_wide1 / k_big2, the host solver,
# searches with and without the tail incl. one- and two-layer models and the one-launch residual steps)
DGCN_FUZZ_CASES=1000 timeout 3300 python -m pytest tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r05_fuzz_1000.log 2>&1
tail -12 gpurun_out/r05_fuzz_1000.log
