#!/bin/bash
# round 5, call 24: residual steps on 977 .. 1 920 vertices in one launch of k_big2<RESID>
timeout 1500 python -m pytest tests/test_gpu_big2.py -x -q --tb=short -p no:cacheprovider 2>&1 | tail -4
for v in 1 0; do echo "DGCN_BIG_RESIDUAL=$v"; DGCN_BIG_RESIDUAL=$v python tools/run_iterative.py --graphs 64 --family mc --n 1500 --p 0.03 --layers 20 --host 0 2>/dev/null | grep -v path | cut -c1-260; done
