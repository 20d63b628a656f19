#!/bin/bash
# round 4, call 41: k_big record build with all of a wave's tiles in flight
for c in er500 mc900; do for l in 3 20; do
  python tools/run_general.py $c 200 $l 256 2>/dev/null | grep big_solve
done; done
python tools/run_general.py er200x0.1 200 20 500 2>/dev/null | grep big_
timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider 2>&1 | tail -2
