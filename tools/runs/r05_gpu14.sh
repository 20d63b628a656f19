#!/bin/bash
# round 5, call 14: k_fused C3 - the two workgroups of a CU take turns in the aggregation phase (DGCN_FUSED_TURN)
for v in 0 1 0 1; do echo -n "DGCN_FUSED_TURN=$v: "; DGCN_FUSED_TURN=$v python tools/run_fused.py er 300 20 500 2>/dev/null | tail -1; done
for v in 0 1; do echo -n "DGCN_FUSED_TURN=$v BA mix: "; DGCN_FUSED_TURN=$v python tools/run_fused.py ba 300 20 500 2>/dev/null | tail -1; done
for v in 0 1; do echo -n "DGCN_FUSED_TURN=$v 400 graphs: "; DGCN_FUSED_TURN=$v python tools/run_fused.py er 300 20 400 2>/dev/null | tail -1; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "full_size or fused" 2>&1 | tail -2
