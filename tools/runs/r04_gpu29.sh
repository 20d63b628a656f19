#!/bin/bash
# round 4, call 29: the tail kernel (DGCN_RESIDUAL_FINISH_SMALL): its tests, the residual-step tests of the other suites, C5 and iterative timings
timeout 1500 python -m pytest tests/test_gpu_tail.py tests/test_gpu_general.py tests/test_gpu_api.py -x -q --tb=short -p no:cacheprovider -k "tail or finish or residual or iterative or wireless or rollout or dit or cit" > gpurun_out/r04_gpu29.log 2>&1
tail -30 gpurun_out/r04_gpu29.log
python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > gpurun_out/r04_gpu29_c5.txt
python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > gpurun_out/r04_gpu29_mc.txt
cat gpurun_out/r04_gpu29_c5.txt gpurun_out/r04_gpu29_mc.txt
