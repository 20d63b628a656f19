#!/bin/bash
# round 6, call 39: the C ABI from six host threads at once, each on its own stream
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -15; done | tee gpurun_out/r06_concurrency.txt
