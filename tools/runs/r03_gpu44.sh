#!/bin/bash
# round 3, call 44: record-slice guard (a row longer than its graph), whole GPU suite
python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
