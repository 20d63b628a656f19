#!/bin/bash
# round 5, call 16: k_res_cand as per-wave selection + merge: the rollout tests, then searches on the any-size path
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_wide.py tests/test_gpu_big2.py -x -q --tb=short -p no:cacheprovider -k "rollout or residual or iterative or witness or agree" 2>&1 | tail -4
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-600
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 1 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-600
python tools/run_iterative.py --graphs 64 --n 500 --p 0.1 --layers 20 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-600
python tools/run_iterative.py --graphs 8 --n 5000 --p 0.001 --layers 1 --host 0 --only rollout 2>/dev/null | grep -v path | cut -c1-600
