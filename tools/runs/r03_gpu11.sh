#!/bin/bash
# round 3, call 11: e2e with the expansion on the copy stream (A/B), full GPU suite, then the round's profile collection
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact   :', d['e2e']['value'], d['e2e']['ms_per_batch'], d['ms_per_step'])"
DGCN_HOST_COMPACT=0 python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no compact:', d['e2e']['value'], d['e2e']['ms_per_batch'])"
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact   :', d['e2e']['value'], d['e2e']['ms_per_batch'], d['ms_per_step'])"
python -m pytest tests -m gpu -q 2>&1 | tail -6
bash tools/collect_profiles_r03.sh 2>&1 | tail -25
