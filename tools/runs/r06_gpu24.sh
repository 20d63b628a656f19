#!/bin/bash
# round 6, call 24: HBM traffic of k_fused<.., GW> on C4 (4 000 BA graphs) - FETCH_SIZE / WRITE_SIZE passes (separate runs, --pmc with --kernel-trace only)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_gw; rm -rf $O; mkdir -p $O; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/gw_$c -- python3 $R/tools/run_fused.py ba 5 20 4000 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob("gpurun_out/pmc_gw/gw_%s/**/*counter_collection.csv" % c, recursive=True), key=os.path.getmtime)
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"] and r["Counter_Name"] == c:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    print(c, sum(per.values()) / len(per), len(per), set(r["Kernel_Name"][:60] for r in csv.DictReader(open(f)) if "k_fused" in r["Kernel_Name"]))
PY
