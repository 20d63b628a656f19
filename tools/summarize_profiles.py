#!/usr/bin/env python3
"""Turn gpurun_out/profiles/* (tools/collect_profiles.sh) into the tracked summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(DST, exist_ok=True)


def first(pattern):
    """Newest match: gpurun MERGES each call's output into gpurun_out/, so older runs linger next to it."""
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


for mode in ("fused", "layered"):
    f = first(os.path.join(SRC, "bench_" + mode, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(DST, "%s_bench_%s_kernel_stats.csv" % (TAG, mode)))
    j = os.path.join(SRC, "bench_%s.json" % mode)
    if os.path.isfile(j):
        lines = [l for l in open(j) if l.startswith("{")]
        if lines:
            open(os.path.join(DST, "%s_bench_%s.json" % (TAG, mode)), "w").write(lines[-1])


f = first(os.path.join(SRC, "iterative", "**", "*kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(DST, "%s_iterative_n500_kernel_stats.csv" % TAG))
    j = os.path.join(SRC, "iterative.json")
    if os.path.isfile(j):
        open(os.path.join(DST, "%s_iterative_n500.json" % TAG), "w").write("".join(l for l in open(j) if l.startswith("{")))


def pmc(kind, counter, match):
    f = first(os.path.join(SRC, "pmc_%s_%s" % (kind, counter), "**", "*counter_collection.csv"))
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals) if vals else None


traffic = {}
notes = []
for kind, match, key in (("fused", "k_fused", "fused_solve|500x200|l20"), ("spmm", "k_spmm_lds", "spmm|500x200|C32")):
    fe, wr = pmc(kind, "FETCH_SIZE", match), pmc(kind, "WRITE_SIZE", match)
    if fe is None or wr is None:
        continue
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B in rocprofv3's derived
    # metric; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads.
    # Report both the raw and the doubled read figure; 'hbm_bytes_per_launch' uses the doubled one
    # (upper bound for narrower accesses, which the guide calls uncalibrated).
    raw = (fe + wr) * 1024.0
    corrected = (2.0 * fe + wr) * 1024.0
    traffic[key] = {"FETCH_SIZE": fe, "WRITE_SIZE": wr, "raw_bytes_per_launch": raw, "hbm_bytes_per_launch": corrected}
    notes.append("%s: FETCH_SIZE %.1f, WRITE_SIZE %.1f (KiB units) -> %.2f MB raw, %.2f MB with the gfx950 read correction"
                 % (key, fe, wr, raw / 1e6, corrected / 1e6))
json.dump(traffic, open(os.path.join(DST, "hbm_traffic.json"), "w"), indent=1)
open(os.path.join(DST, "%s_hbm_traffic.txt" % TAG), "w").write("\n".join(notes) + "\n")
print("\n".join(notes))
for f in sorted(os.listdir(DST)):
    print(f)
