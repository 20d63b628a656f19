#!/bin/bash
# round 5, call 27: the rollout step's candidates selected inside the step's own launch, the instances' masks made by k_lgs: parity + step times
python -m pytest tests/test_gpu_general.py tests/test_gpu_wide.py tests/test_gpu_big2.py -m gpu -x -q 2>&1 | tail -4
for args in "--family mc --graphs 64 --n 900 --p 0.03 --layers 20" "--family mc --graphs 64 --n 900 --p 0.03 --layers 1" "--family mc --graphs 64 --n 1500 --p 0.03 --layers 20" "--graphs 64 --n 700 --p 0.02 --layers 20"; do
  python tools/run_iterative.py $args --host 0 --only rollout 2>&1 | grep -v '^{"path'
done
