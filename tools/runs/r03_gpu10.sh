#!/bin/bash
# round 3, call 10: compact transfers (u16 ids + degrees), persistent SpMM with reordered pipeline, shallow block choice; e2e
python -m pytest tests/test_gpu_kernels.py -q -k "shallow or spmm" 2>&1 | tail -3
python -m pytest tests/test_gpu_api.py -q -k "host_solver or serving or compact" 2>&1 | tail -3
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500"; do python tools/run_fused.py $cfg; done
echo "== spmm"; python tools/tune_spmm_hbm.py 2>&1 | grep -v "GLOBAL\|CSRCAP" | tail -2; DGCN_SPMM_PERSIST=0 python tools/tune_spmm_hbm.py 2>&1 | grep -v "GLOBAL\|CSRCAP" | tail -2
echo "== e2e"
python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact   :', d['e2e']['value'], d['e2e']['ms_per_batch'], d['ms_per_step'])"
DGCN_HOST_COMPACT=0 python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no compact:', d['e2e']['value'], d['e2e']['ms_per_batch'])"
python tools/pack_probe.py 8; python tools/pack_probe.py 16
