#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -k "native_host_solver or solve_mwis or heuristics or wireless" 2>&1 | tail -3
echo "== spin"; python tools/run_single.py 400 2>&1 | tail -4
echo "== block"; DGCN_HOST_SPIN=0 python tools/run_single.py 400 2>&1 | tail -4
echo "== spin, K=6"; DGCN_FUSED_CLUSTER=6 python tools/run_single.py 400 2>&1 | tail -4
