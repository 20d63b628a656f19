#!/bin/bash
# round 5, call 36: the whole GPU suite on the final kernels, then the round's evidence (tools/collect_profiles_r05.sh)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05_gpu_suite.txt
cat gpurun_out/r05_gpu_suite.txt
timeout 3000 bash tools/collect_profiles_r05.sh 2>&1 | tail -5
