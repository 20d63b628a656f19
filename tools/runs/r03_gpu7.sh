#!/bin/bash
# round 3, call 7: shallow kernel v2 (one barrier per round, register-cached ids), persistent SpMM, even row deal A/B
python -m pytest tests/test_gpu_kernels.py -q -k "shallow or ties or spmm or precise or golden" 2>&1 | tail -4
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500"; do python tools/run_fused.py $cfg; done
bash tools/build_diag.sh 2>&1 | tail -2
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shallow_stamps3.txt
echo "== C3 even rows A/B"; for i in 1 2; do python tools/run_fused.py er 300 20 500; DGCN_FUSED_EVEN=1 python tools/run_fused.py er 300 20 500; done
DGCN_FUSED_EVEN=1 python tools/run_fused.py ba 100 20 500; python tools/run_fused.py ba 100 20 500
echo "== spmm"; python tools/tune_spmm_hbm.py 2>&1 | tail -12; DGCN_SPMM_PERSIST=0 python tools/tune_spmm_hbm.py 2>&1 | tail -12
