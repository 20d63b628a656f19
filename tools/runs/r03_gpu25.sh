#!/bin/bash
# round 3, call 25: which of the shallow kernel's changes costs the C2 launch its microsecond (variants built with -D switches)
for i in 1 2 3; do
for lib in head "" v2 v3; do
f=distgcn_amd/libdgcn${lib:+_$lib}.so
DGCN_LIB=$f python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
