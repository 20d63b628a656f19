#!/usr/bin/env python3
"""gpurun_out/profiles_r06/* (tools/collect_profiles_r06.sh) -> tracked summaries under profiles/ (prefix r06_)."""
import csv, glob, json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r06")
DST = os.path.join(ROOT, "profiles")
TAG = "r06"


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def keep(name):
    f = newest(os.path.join(SRC, name, "**", "*kernel_stats.csv"))
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(DST, "%s_%s_kernel_stats.csv" % (TAG, name)), "w", newline="") as out:
            csv.writer(out).writerows(rows[:13])  # the file stays a summary: top 12 kernels
    for suffix, tail in ((".json", "_profiled_run.json"), (".unprofiled.json", ".json")):
        j = os.path.join(SRC, name + suffix)
        if os.path.isfile(j):
            lines = [l for l in open(j) if l.startswith("{")]
            if lines:
                open(os.path.join(DST, "%s_%s%s" % (TAG, name, tail)), "w").write(lines[-1])


NAMES = ("bench_default", "bench_mc900_l1", "bench_mc1500", "bench_mc900", "bench_er500", "bench_c2", "bench_c4_l1", "bench_c4_l20", "bench_c5",
         "bench_c5_256", "bench_c5_512", "bench_mc900_rollout", "bench_mc900_rollout_l1", "iterative_mc900")
for name in NAMES:
    keep(name)
for f in ("iterative_mc900.txt", "iterative_mc900_l1.txt", "iterative_mc1500.txt", "iterative_er500.txt", "iterative_c5.txt",
          "wireless_mc900_l1.txt", "wireless_mc900_l20.txt", "big2_phase_clocks.txt", "big_phase_clocks.txt", "fused_phase_clocks.txt",
          "narrow_and_poly.txt"):
    if os.path.isfile(os.path.join(SRC, f)):
        open(os.path.join(DST, "%s_%s" % (TAG, f)), "w").write(open(os.path.join(SRC, f)).read())


def pmc(kind, counter, match):
    f = newest(os.path.join(SRC, "pmc_%s_%s" % (kind, counter), "**", "*counter_collection.csv"))
    if not f:
        return None
    per = {}
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return sum(per.values()) / len(per) if per else None


tpath = os.path.join(DST, "hbm_traffic.json")
traffic = json.load(open(tpath)) if os.path.isfile(tpath) else {}
notes = []
for kind, match, key in (("er500", "k_big", "big_solve|256x500|l20"), ("mc900", "k_big", "big_solve|mc256x900|l20"),
                         ("mc1500", "k_big2", "big_solve|mc256x1500|l20"), ("mc900l1", "k_wide1", "wide_solve|mc256x900|l1"),
                         ("c3", "k_fused", "fused_solve|500x200|l20"), ("c5", "k_fused", "fused_residual|64x500|l20"),
                         ("mc900roll", "k_big", "big_residual|64x900|l20"), ("mc900rolll1", "k_wide1", "wide_residual|64x900|l1")):
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
json.dump(traffic, open(tpath, "w"), indent=1)
open(os.path.join(DST, "%s_hbm_traffic.txt" % TAG), "w").write("\n".join(notes) + "\n")
print("\n".join(notes))
for name in NAMES:
    p = os.path.join(DST, "%s_%s.json" % (TAG, name))
    if os.path.isfile(p):
        d = json.loads(open(p).read())
        r = d.get("roofline") or {}
        print("%-14s value %12.1f  ms/step %.4f  kernel %.1f us  frac %s  cpu %s" % (
            name, d["value"], d["ms_per_step"], r.get("avg_launch_us") or 0, r.get("frac"), (d.get("cpu_baseline") or {}).get("value")))
