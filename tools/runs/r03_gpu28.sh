#!/bin/bash
# round 3, call 28: shallow kernels chosen by density; phase clocks of k_fused on the BA mix (where does the hub block leave wave 0?)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "shallow or fuzz or ties" 2>&1 | tail -2
for lib in head ""; do f=distgcn_amd/libdgcn${lib:+_$lib}.so; DGCN_LIB=$f python tools/run_fused.py er200 300 1 500; DGCN_LIB=$f python tools/run_fused.py ba 300 1 500; DGCN_LIB=$f python tools/run_fused.py er100 300 1 500; done
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py ba 20 500 2>&1 | grep -v amdgpu.ids | head -40
python bench.py --config C4-share --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4-l20:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
