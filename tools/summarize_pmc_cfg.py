#!/usr/bin/env python3
"""gpurun_out/pmc_<cfg>/* (tools/collect_pmc_cfg.sh) -> profiles/<tag>_<cfg>_pmc.txt: per-launch sums of each counter for
the solve kernel of that configuration.  python tools/summarize_pmc_cfg.py <tag> <cfg> "<description>" [kernel substring]"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg, desc = sys.argv[1], sys.argv[2], sys.argv[3]
kern = sys.argv[4] if len(sys.argv) > 4 else "k_"
sub = os.path.join(ROOT, "gpurun_out", "pmc_" + cfg)
acc, cnt, names = collections.defaultdict(float), collections.defaultdict(set), collections.Counter()
groups = collections.defaultdict(list)
for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
    groups[os.path.relpath(f, sub).split(os.sep)[0]].append(f)
for f in (max(fs, key=os.path.getmtime) for fs in groups.values()):
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"] or "k_graph_rank" in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
        names[r["Kernel_Name"].split("(")[0]] += 1
lines = ["%s: counter sums per launch (all XCDs / SEs / CUs); kernel(s): %s" % (desc, ", ".join(sorted(names)))]
for k in sorted(acc):
    lines.append("%-32s %16.0f" % (k, acc[k] / max(len(cnt[k]), 1)))
if "FETCH_SIZE" in acc or "WRITE_SIZE" in acc:
    f, w = acc.get("FETCH_SIZE", 0) / max(len(cnt["FETCH_SIZE"]), 1), acc.get("WRITE_SIZE", 0) / max(len(cnt["WRITE_SIZE"]), 1)
    lines.append("HBM traffic per launch (rocprofv3 reports KiB; MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts half the "
                 "bytes of wide coalesced reads - doubled here): read %.2f MB, written %.2f MB, total %.2f MB"
                 % (2 * f * 1024 / 1e6, w * 1024 / 1e6, (2 * f + w) * 1024 / 1e6))
out = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (tag, cfg))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
