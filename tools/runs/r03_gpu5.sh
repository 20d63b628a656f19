#!/bin/bash
# round 3, call 5: the one-layer kernel (shallow.hip): parity tests, fuzz, timings against k_fused on C2 / C4 l=1
python -m pytest tests/test_gpu_kernels.py -q -k "shallow or ties or golden or empty" 2>&1 | tail -5
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500" "er100 300 1 4000"; do
  python tools/run_fused.py $cfg; DGCN_SHALLOW=0 python tools/run_fused.py $cfg
done
python bench.py --config C2 --steps 2000 --no-e2e --no-spmm-probe --no-cpu-pool --cpu-seconds 3 2>&1 | tail -1 > gpurun_out/r03_gpu5_c2.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_gpu5_c2.json').read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["parity_full_size"], d["margin_risk"])
PY
