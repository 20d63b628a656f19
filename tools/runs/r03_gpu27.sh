#!/bin/bash
# round 3, call 27: long-row threshold at 16 lane trips
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "shallow or fuzz or ties" 2>&1 | tail -2
for i in 1 2; do
for lib in head ""; do f=distgcn_amd/libdgcn${lib:+_$lib}.so; DGCN_LIB=$f python tools/run_fused.py er200 300 1 500; DGCN_LIB=$f python tools/run_fused.py ba 300 1 500; DGCN_LIB=$f python tools/run_fused.py er100 300 1 500; done
done
for lib in head ""; do
f=distgcn_amd/libdgcn${lib:+_$lib}.so
DGCN_LIB=$f python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
