#!/bin/bash
# round 4, call 50: k_big tile-to-wave mapping reversed (3) against the plain mapping (0)
for p in 0 4; do for c in er500 mc900; do
  echo -n "DGCN_BIG_PRIO=$p "; DGCN_BIG_PRIO=$p python tools/run_general.py $c 300 20 256 2>/dev/null | grep big_solve
done; done
DGCN_BIG_PRIO=4 timeout 600 python -m pytest tests/test_gpu_general.py -x -q -p no:cacheprovider -k "big_graphs or plain" 2>&1 | tail -2
