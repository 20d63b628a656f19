#!/bin/bash
# round 3, call 26: shallow kernel final shape (A/B against the committed build), cluster auto rule back to <= 4 tiles, whole GPU suite
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for lib in head ""; do
f=distgcn_amd/libdgcn${lib:+_$lib}.so
DGCN_LIB=$f python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
DGCN_LIB=$f python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
for lib in head ""; do f=distgcn_amd/libdgcn${lib:+_$lib}.so; DGCN_LIB=$f python tools/run_fused.py er200 300 1 500; DGCN_LIB=$f python tools/run_fused.py ba 300 1 500; done
python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5:', d['value'], d['ms_per_step'])"
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shallow_stamps8.txt
