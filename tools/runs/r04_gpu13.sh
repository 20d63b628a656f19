#!/bin/bash
# round 4, call 13: the iterative solvers on graphs beyond the fused kernel (any-size path): 64 joint 3 x 300 graphs, 64 ER(500, 0.1)
mkdir -p gpurun_out
python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 2>&1 | grep "^{" > gpurun_out/r04_iterative_mc900.txt
python tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 2>&1 | grep "^{" > gpurun_out/r04_iterative_er500.txt
python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>&1 | grep "^{" > gpurun_out/r04_iterative_c5.txt
DGCN_GENERAL=1 python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>&1 | grep "^{" > gpurun_out/r04_iterative_c5_general.txt
cat gpurun_out/r04_iterative_*.txt
