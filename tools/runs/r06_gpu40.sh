#!/bin/bash
# round 6, call 40: the driver's own command line (--steps 20 --warmup 5) three times un-profiled, then once under rocprofv3 --kernel-trace:
# how do the 20 timed launches compare with the steady state of the default run?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/driver_cmd
rm -rf $O; mkdir -p $O
cd /tmp
for i in 1 2 3 4 5; do python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --parity-seconds 0 --no-spmm-probe --cpu-seconds 0 2>/dev/null | tail -1 > $O/run$i.json; done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --parity-seconds 0 --no-spmm-probe --cpu-seconds 0 > $O/traced.json 2> $O/traced.err
python3 - <<PY
import json, glob, csv
for f in sorted(glob.glob("$O/run*.json")) + ["$O/traced.json"]:
    d = json.load(open(f)); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"]*1e3, 2), round(d["roofline"].get("avg_launch_us", 0), 2), d.get("without_launch_events", {}).get("ms_per_step"))
rows = []
for f in glob.glob("$O/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
print("k_fused launches:", len(rows))
tail = rows[-60:]
prev = None
for s, e in tail:
    print("%8.1f us  gap before %8.1f us" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0)); prev = e
PY
