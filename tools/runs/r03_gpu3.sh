#!/bin/bash
# round 3, call 3: the GPU suite and a C3 bench step with the new arithmetic contract (layer 0 aggregation / layer 1 transform in double)
python -m pytest tests -m gpu -q --maxfail=25 -x --deselect tests/test_gpu_fuzz.py 2>&1 | tail -40 > gpurun_out/r03_gpu3_pytest.txt
tail -15 gpurun_out/r03_gpu3_pytest.txt
python bench.py --steps 400 --cpu-seconds 0 --no-e2e --no-spmm-probe --no-cpu-pool 2>&1 | tail -1 > gpurun_out/r03_gpu3_bench.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_gpu3_bench.json').read()); print(d["value"], d["ms_per_step"], d["roofline"])
PY
python tools/run_fused.py er100 200 1 500; python tools/run_fused.py ba 200 1 500; python tools/run_fused.py ba 100 20 500; python tools/run_fused.py er 100 20 1
