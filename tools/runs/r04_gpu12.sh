#!/bin/bash
# round 4, call 12: the round's evidence run - bench lines (un-profiled + kernel trace) for the default command, ER500, MC900 and
# the BASELINE configs, PMC traffic of k_big; then the full GPU suite
bash tools/collect_profiles_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu12.log 2>&1
tail -4 gpurun_out/r04_gpu12.log
