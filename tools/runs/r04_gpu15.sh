#!/bin/bash
# round 4, call 15: k_res_cand rewritten (lanes per vertex, loads in flight), 1024-thread residual kernels: tests + iterative timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04_gpu15.log 2>&1
tail -3 gpurun_out/r04_gpu15.log
python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 2>&1 | grep "^{" > gpurun_out/r04_iterative2_mc900.txt
python tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 --host 0 2>&1 | grep "^{" > gpurun_out/r04_iterative2_er500.txt
cat gpurun_out/r04_iterative2_*.txt
bash tools/runs/r04_gpu14.sh
