#!/bin/bash
# round 5, call 6: k_big2 with real branches round the tile register picks
for c in mc1500 er1500x0.004 er1000x0.01; do python tools/run_general.py $c 100 20 256 2>/dev/null | grep -v path; done
timeout 600 python -m pytest tests/test_gpu_big2.py -x -q -p no:cacheprovider -k "plain_solve" 2>&1 | tail -2
