#!/bin/bash
# round 3, call 24: shallow kernel LONG variant only above 128 vertices (A/B against the committed build); cluster variant
# with up to eight tiles per workgroup (C5: N = 500 graphs on four CUs each)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for lib in distgcn_amd/libdgcn_head.so distgcn_amd/libdgcn.so; do
DGCN_LIB=$lib python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
for lib in distgcn_amd/libdgcn_head.so distgcn_amd/libdgcn.so; do
DGCN_LIB=$lib python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
DGCN_LIB=$lib python tools/run_fused.py er200 300 1 500
DGCN_LIB=$lib python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220
DGCN_LIB=$lib python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib C5:', d['value'], d['ms_per_step'])"
done
DGCN_FUSED_CLUSTER=0 python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220
