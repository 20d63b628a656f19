#!/usr/bin/env python3
"""gpurun_out/pmc/* (tools/collect_pmc.sh) -> profiles/<tag>_fused_pmc.txt: per-launch sums of each counter for k_fused."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
GRAPHS = sys.argv[2] if len(sys.argv) > 2 else "500"  # "1": gpurun_out/pmc_1 -> profiles/<tag>_cluster_pmc.txt
SUB = "pmc" if GRAPHS == "500" else "pmc_" + GRAPHS
acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
# gpurun MERGES every call's output into gpurun_out/, so files of earlier rounds linger: newest file per counter group
groups = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", SUB, "**", "*counter_collection.csv"), recursive=True):
    groups[os.path.relpath(f, os.path.join(ROOT, "gpurun_out", SUB)).split(os.sep)[0]].append(f)
for f in (max(fs, key=os.path.getmtime) for fs in groups.values()):
    for r in csv.DictReader(open(f)):
        if "k_fused" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
lines = ["k_fused<false,false,512%s>, %s ER graph(s) N=200, l=20 (tools/run_fused.py er 5 20 %s under tools/collect_pmc.sh): counter sums per launch (all XCDs / SEs / CUs)"
         % (",cluster" if GRAPHS != "500" else "", GRAPHS, GRAPHS)]
for k in sorted(acc):
    lines.append("%-32s %16.0f" % (k, acc[k] / max(len(cnt[k]), 1)))
open(os.path.join(ROOT, "profiles", ("%s_fused_pmc.txt" if GRAPHS == "500" else "%s_cluster_pmc.txt") % TAG), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
