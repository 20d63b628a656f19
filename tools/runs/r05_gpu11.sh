#!/bin/bash
# round 5, call 11: residual steps of deep models in one launch of k_big: tests, then MC900 / ER500 searches with and without
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_tail.py tests/test_gpu_fuzz.py -x -q --tb=short -p no:cacheprovider 2>&1 | tail -6
for v in 1 0; do for cfg in "--family mc --n 900 --p 0.03" "--n 500 --p 0.1"; do echo "DGCN_BIG_RESIDUAL=$v $cfg"; DGCN_BIG_RESIDUAL=$v python tools/run_iterative.py --graphs 64 $cfg --layers 20 --host 0 2>/dev/null | grep -v path; done; done
for c in er500 mc900; do python tools/run_general.py $c 300 20 256 2>/dev/null | grep big_solve; done
