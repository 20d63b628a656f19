#!/bin/bash
# round 3, call 23: shallow kernel A/B on one box, votes read behind the register path
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "shallow or fuzz or ties" 2>&1 | tail -2
for i in 1 2 3 4; do
for lib in distgcn_amd/libdgcn_head.so distgcn_amd/libdgcn.so; do
DGCN_LIB=$lib python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
DGCN_LIB=distgcn_amd/libdgcn.so python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
