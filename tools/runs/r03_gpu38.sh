#!/bin/bash
# round 3, call 38: shallow kernel - priority and stamp of a vertex in one 16-byte LDS word (one read per neighbour): A/B on one box
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "shallow or fuzz or ties" 2>&1 | tail -2
for i in 1 2 3; do
for lib in prev ""; do
f=distgcn_amd/libdgcn${lib:+_$lib}.so
DGCN_LIB=$f python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
DGCN_LIB=$f python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
for lib in prev ""; do f=distgcn_amd/libdgcn${lib:+_$lib}.so; DGCN_LIB=$f python tools/run_fused.py er200 300 1 500 2>/dev/null; done
