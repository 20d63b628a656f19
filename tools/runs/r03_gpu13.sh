#!/bin/bash
# round 3, call 13: A/B of rotating the second workgroup's left-over tiles / row blocks (DGCN_FUSED_ROT)
for i in 1 2 3; do python tools/run_fused.py er 400 20 500; DGCN_FUSED_ROT=1 python tools/run_fused.py er 400 20 500; done
python -m pytest tests/test_gpu_kernels.py -q -k "solve_full_size or golden or ties" 2>&1 | tail -2
DGCN_FUSED_ROT=1 python -m pytest tests/test_gpu_kernels.py -q -k "solve_full_size or golden or ties" 2>&1 | tail -2
