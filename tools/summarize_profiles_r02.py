#!/usr/bin/env python3
"""gpurun_out/profiles_r02/* (tools/collect_profiles_r02.sh) -> tracked summaries under profiles/ (prefix r02_)."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r02")
DST = os.path.join(ROOT, "profiles")
TAG = "r02"


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def keep_stats(name):
    f = newest(os.path.join(SRC, name, "**", "*kernel_stats.csv"))
    if f:
        rows = list(csv.reader(open(f)))
        # drop torch's own helper kernels beyond the top 12 rows: the file stays a summary
        with open(os.path.join(DST, "%s_%s_kernel_stats.csv" % (TAG, name)), "w", newline="") as out:
            csv.writer(out).writerows(rows[:13])
    j = os.path.join(SRC, name + ".json")
    if os.path.isfile(j):
        lines = [l for l in open(j) if l.startswith("{") or l.startswith("fused_solve")]
        if lines:
            open(os.path.join(DST, "%s_%s.json" % (TAG, name)), "w").write("".join(lines))


for name in ("bench_fused", "bench_layered", "bench_c2", "bench_c4_l1", "bench_c4_l20", "c5_iterative", "spmm_cache", "spmm_rot8",
             "spmm_one4000", "bench_two_streams"):
    keep_stats(name)


def pmc(kind, counter, match):
    f = newest(os.path.join(SRC, "pmc_%s_%s" % (kind, counter), "**", "*counter_collection.csv"))
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals) if vals else None


traffic, notes = {}, []
for kind, match, key in (("fused", "k_fused", "fused_solve|500x200|l20"), ("spmm", "k_spmm_lds", "spmm|500x200|C32"),
                         ("spmmrot8", "k_spmm_lds", "spmm|500x200|C32|out_of_cache"),
                         ("spmmone4000", "k_spmm_lds", "spmm|500x200|C32|one4000")):
    fe, wr = pmc(kind, "FETCH_SIZE", match), pmc(kind, "WRITE_SIZE", match)
    if fe is None or wr is None:
        continue
    # MI355X_MICROARCH.md (HBM): rocprofv3's FETCH_SIZE / WRITE_SIZE are in units of 1 KiB; on gfx950 FETCH_SIZE reports half
    # the bytes of wide (16 B per lane) coalesced streaming reads - double it; WRITE_SIZE is exact for 16-B streaming stores.
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
