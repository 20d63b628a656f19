#!/bin/bash
# round 4, call 5: k_big (hidden stack in one launch) inside the any-size path: tests, then the first-measurement shapes again
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04_gpu11.log 2>&1
tail -5 gpurun_out/r04_gpu11.log
for cfg in "256 500 0.1" "128 900 0.01" "256 900 0.01" "500 200 0.1"; do
  set -- $cfg
  DGCN_GENERAL=1 python bench.py --graphs $1 --nodes $2 --p $3 --layers 20 --steps 40 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool 2>/dev/null | tail -1 > gpurun_out/r04_big5_general_$1x$2.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_big5_*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("kernel_us"))
    except Exception as e: print(f, "ERR", e, open(f).read()[:300])
PY
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_prof6
rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/er500" -- python3 $R/bench.py --graphs 256 --nodes 500 --p 0.1 --layers 20 --steps 100 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool > "$O/er500.json" 2> "$O/er500.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/n900" -- python3 $R/bench.py --graphs 256 --nodes 900 --p 0.01 --layers 20 --steps 100 --warmup 5 --no-e2e --no-spmm-probe --cpu-seconds 0 --parity-seconds 0 --no-cpu-pool > "$O/n900.json" 2> "$O/n900.err"
cd $R
for d in er500 n900; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
