#!/bin/bash
# round 4, call 3: first measurements of the any-size path (layer by layer inside dgcn_solve_batch): ER(500, 0.1), N = 900, N = 1500
mkdir -p gpurun_out
for cfg in "256 500 0.1" "128 900 0.01" "128 1500 0.004" "500 200 0.1"; do
  set -- $cfg
  python bench.py --graphs $1 --nodes $2 --p $3 --layers 20 --steps 40 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool 2>/dev/null | tail -1 > gpurun_out/r04_first_$1x$2.json
  DGCN_GENERAL=1 python bench.py --graphs $1 --nodes $2 --p $3 --layers 20 --steps 40 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool 2>/dev/null | tail -1 > gpurun_out/r04_first_general_$1x$2.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_first_*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], {k: (round(v["avg_us"],1), round(v["launches_per_step"],1)) for k,v in d.get("kernel_us",{}).items()})
    except Exception as e: print(f, "ERR", e, open(f).read()[:300])
PY
