#!/bin/bash
# round 3, call 18: records requested two trips ahead; phase ablation of the C3 launch (diag build)
for i in 1 2; do
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/ablate_fused.py 2>&1 | grep -v amdgpu.ids
