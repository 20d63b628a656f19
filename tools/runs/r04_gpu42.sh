#!/bin/bash
# round 4, call 42: evidence refresh on the final kernels (tail kernel, mask search in k_big): bench lines, kernel traces, PMC traffic
# of k_big, iterative timings; smoke(); the full GPU suite
bash tools/collect_profiles_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu42.log 2>&1; tail -3 gpurun_out/r04_gpu42.log
