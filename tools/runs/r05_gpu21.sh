#!/bin/bash
# round 5, call 21: k_res_pick across lanes; rollout tests; searches
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_wide.py tests/test_gpu_big2.py -x -q --tb=short -p no:cacheprovider -k "rollout or residual" 2>&1 | tail -3
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 2>/dev/null | grep -v path | cut -c1-420
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 1 --host 0 2>/dev/null | grep -v path | cut -c1-420
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 2 --host 0 2>/dev/null | grep -v path | cut -c1-420
