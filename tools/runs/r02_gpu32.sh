#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
echo "== library order on (default)"; python tools/order_probe.py | tail -6
echo "== library order off"; DGCN_FUSED_ORDER=0 python tools/order_probe.py | tail -6
echo "== C3 with order forced on / default"; DGCN_FUSED_ORDER=1 python tools/time_fused.py forced; python tools/time_fused.py default
