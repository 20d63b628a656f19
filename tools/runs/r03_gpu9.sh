#!/bin/bash
# round 3, call 9: shallow v2 restored + block-size sweep; persistent SpMM (selection fixed); compact transfers; e2e
python -m pytest tests/test_gpu_kernels.py -q -k "shallow or ties or spmm or precise" 2>&1 | tail -3
python -m pytest tests/test_gpu_api.py -q -k "host_solver or serving or compact" 2>&1 | tail -3
for blk in 128 256 512 1024; do echo "== shallow block $blk"; for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500"; do DGCN_SHALLOW_BLOCK=$blk python tools/run_fused.py $cfg; done; done
echo "== spmm"; python tools/tune_spmm_hbm.py 2>&1 | grep -v GLOBAL | tail -4; DGCN_SPMM_PERSIST=0 python tools/tune_spmm_hbm.py 2>&1 | grep -v GLOBAL | tail -4
echo "== bench default (e2e with compact transfers)"
python bench.py --cpu-seconds 3 --no-cpu-pool 2>&1 | tail -1 > gpurun_out/r03_gpu9_bench.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_gpu9_bench.json').read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]); print(d["e2e"]); print(d["parity_full_size"]); print(d["margin_risk"]); print(d["single_graph"])
PY
DGCN_HOST_COMPACT=0 python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no compact:', d['e2e'])"
