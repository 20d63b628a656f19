#!/bin/bash
# round 3, call 12: nonce-stamped cluster flags (no memset), f64 MFMA transform + LDS-staged f64 SpMM in the layered path
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --mode layered --steps 200 --warmup 5 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('layered:', d['ms_per_step'], d['kernels'])"
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], 'single', d['single_graph']['call_us'], d['single_graph']['kernel_us'], 'e2e', d['e2e']['value'])"
python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], 'single', d['single_graph']['call_us'], d['single_graph']['kernel_us'], 'e2e', d['e2e']['value'])"
python bench.py --config C4-share --layers 1 --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4-l1:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
python tools/run_single.py 400 2>&1 | tail -4
python tools/time_small.py 1 2>&1 | tail -3
