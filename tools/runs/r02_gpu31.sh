#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -k "native_host_solver or solve_mwis or heuristics or wireless or dqn or gdpg" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cluster" 2>&1 | tail -2
echo "== word"; python tools/lat_probe.py 1000 2>&1 | tail -4
echo "== event"; DGCN_HOST_DONE_WORD=0 python tools/lat_probe.py 1000 2>&1 | tail -4
