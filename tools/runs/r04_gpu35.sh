#!/bin/bash
# round 4, call 35: iterative solvers with / without the tail kernel (DGCN_TAIL=0 switches it off), best of three runs
O=gpurun_out/profiles_r04b; mkdir -p $O
for t in 1 0; do
  echo "DGCN_TAIL=$t"
  DGCN_TAIL=$t python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{"
  DGCN_TAIL=$t python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{"
  DGCN_TAIL=$t python tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{"
done > $O/iterative_tail_on_off.txt
cat $O/iterative_tail_on_off.txt | cut -c1-170
