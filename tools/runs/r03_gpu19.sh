#!/bin/bash
# round 3, call 19: shallow kernel - long rows (BA hubs) entry-parallel
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500"; do python tools/run_fused.py $cfg; done
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shallow_stamps5.txt
python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
