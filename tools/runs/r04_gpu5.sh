#!/bin/bash
# round 4, call 5: k_big (hidden stack in one launch) inside the any-size path: tests, then the first-measurement shapes again
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04_gpu7.log 2>&1
tail -5 gpurun_out/r04_gpu7.log
for cfg in "256 500 0.1" "128 900 0.01" "256 900 0.01" "500 200 0.1"; do
  set -- $cfg
  DGCN_GENERAL=1 python bench.py --graphs $1 --nodes $2 --p $3 --layers 20 --steps 40 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool 2>/dev/null | tail -1 > gpurun_out/r04_big2_general_$1x$2.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_big2_*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("kernel_us"))
    except Exception as e: print(f, "ERR", e, open(f).read()[:300])
PY
