#!/bin/bash
# round 4, call 18: evidence run with the one-launch any-size solve: bench lines (un-profiled + kernel trace), PMC traffic of k_big,
# C5 at 64 / 256 graphs, C3 forced down the any-size path, iterative solvers on MC900 / ER500
bash tools/collect_profiles_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
