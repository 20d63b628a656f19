#!/bin/bash
# round 3, call 16: block-major padded entry records in k_fused (uniform trip counts, no per-lane row bookkeeping)
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], 'single', d['single_graph']['call_us'], d['single_graph']['kernel_us'], 'e2e', d['e2e']['value'])"
done
python bench.py --config C4-share --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4-l20:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5:', d['value'], d['ms_per_step'], d.get('solvers'))"
python tools/time_small.py 20 2>&1 | tail -3
