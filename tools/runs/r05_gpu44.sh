#!/bin/bash
# round 5, call 44: final build - the whole GPU suite, the round's evidence (tools/collect_profiles_r05.sh), the rollout bench lines, the fuzz tests at 1 000 cases
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05_gpu_suite.txt
cat gpurun_out/r05_gpu_suite.txt
timeout 3000 bash tools/collect_profiles_r05.sh 2>&1 | tail -2
python bench.py --config MC900-rollout --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/r05_bench_mc900_rollout.json
python bench.py --config MC900-rollout --layers 1 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r05_bench_mc900_rollout_l1.json
DGCN_FUZZ_CASES=1000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider > gpurun_out/r05_fuzz_1000.log 2>&1
tail -2 gpurun_out/r05_fuzz_1000.log
