#!/bin/bash
# round 3, call 34: evidence pass on the final kernels (tools/collect_profiles_r03.sh) + residual-step phase clocks + fuzz x 200
bash tools/collect_profiles_r03.sh 2>&1 | tail -4
for w in rollout cit; do for b in 0 50; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py $b 64 500 $w 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/profiles_r03/residual_phase_clocks.txt
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220 > gpurun_out/profiles_r03/iterative_steps.txt
DGCN_FUZZ_CASES=200 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2
