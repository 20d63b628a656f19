#!/bin/bash
# round 3, call 4: GPU suite (without -x) + timings after the spill-free rewrite of the precise layers
python -m pytest tests -m gpu -q --deselect tests/test_gpu_fuzz.py 2>&1 | tail -40 > gpurun_out/r03_gpu4_pytest.txt
tail -12 gpurun_out/r03_gpu4_pytest.txt
python bench.py --steps 400 --cpu-seconds 0 --no-e2e --no-spmm-probe --no-cpu-pool 2>&1 | tail -1 > gpurun_out/r03_gpu4_bench.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_gpu4_bench.json').read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
python tools/run_fused.py er100 200 1 500; python tools/run_fused.py ba 200 1 500; python tools/run_fused.py ba 100 20 500; python tools/run_fused.py er 100 20 1
