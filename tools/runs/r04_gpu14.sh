#!/bin/bash
# round 4, call 14: kernel trace of the any-size path's rollout / cit steps on 64 joint 3 x 300 graphs
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_prof_iter
rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/mc900" -- python3 $R/tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 --host 0 > "$O/mc900.txt" 2> "$O/mc900.err"
cd $R
f=$(find $O/mc900 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-64s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
