#!/bin/bash
# round 3, call 6: shallow kernel with coalesced column loads (+ phase clocks), rollout with concurrent completions
python -m pytest tests/test_gpu_kernels.py -q -k "shallow or ties" 2>&1 | tail -3
python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py -q -k "rollout or iterative or residual or c5 or wireless or masked or executed" 2>&1 | tail -5
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500"; do python tools/run_fused.py $cfg; done
bash tools/build_diag.sh 2>&1 | tail -2
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shallow_stamps2.txt
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -4 | cut -c1-200
