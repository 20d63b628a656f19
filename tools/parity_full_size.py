#!/usr/bin/env python3
"""profiles/<tag>_parity_full_size.json: the BASELINE configurations at full size, CPU twin (== HIP bits) against the
oracle restatement.  python tools/parity_full_size.py [tag]   (CPU only, ~1-2 minutes on 8 cores)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import parity
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = {}
for name, (desc, *_rest) in parity.full_size_configs().items():
    t0 = time.time()
    summ, _ = parity.twin_report(name)
    summ["workload"] = desc
    out[name] = summ
    print("%-8s %-34s |hip-f32r| %.2e (%d graphs > 1e-5, %d of them with the restatement further from f64)  |hip-f64| %.2e  "
          "|f32r-f64| %.2e  sets differing %d (unflagged %d)  [%.0fs]"
          % (name, desc, summ["max_err_vs_f32_restatement"], summ["graphs_over_1e-5_vs_f32_restatement"],
             summ["of_those_restatement_further_from_f64"], summ["max_err_vs_f64"], summ["restatement_max_err_vs_f64"],
             summ["sets_differing"], summ["sets_differing_not_flagged_by_margin"], time.time() - t0), flush=True)
with open(os.path.join(ROOT, "profiles", "%s_parity_full_size.json" % tag), "w") as f:
    json.dump(out, f, indent=1)
