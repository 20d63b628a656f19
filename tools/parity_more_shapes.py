#!/usr/bin/env python3
"""The rows of oracle/parity.gpu_extra_configs() (and, with --all, of full_size_configs()) solved on the GPU and held against the
restatement directly: one line per configuration (what tests/test_gpu_full_size.py asserts, as numbers).  Test infrastructure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distgcn_amd.engine import Engine, DeviceModel
from oracle import parity
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_full_size as T


def main():
    eng = Engine("cuda:0")
    rows = dict(T.EXTRA)
    if "--all" in sys.argv:
        rows.update(T.SHARE)
    cfgs = parity.all_configs()
    print("%-16s %6s %9s %5s  %-12s %-12s %-12s %s" % ("configuration", "graphs", "vertices", "path", "max vs f64", "max vs f32", "f32 vs f64", "sets differing / at margin risk"))
    for name in sorted(rows):
        fam, _, model, make = cfgs[name]
        hb = make(rows[name], 0)
        db = eng.upload(hb)
        dm = DeviceModel(parity._model(model), eng.device)
        res = eng.solve_fused(db, dm)
        eng.check_status(res["status"])
        summ, _ = parity.scores_report(name, rows[name], hb.graph_ptr, res["scores"].cpu().numpy().ravel(), res["state"].cpu().numpy())
        print("%-16s %6d %9d %5d  %-12.3g %-12.3g %-12.3g %d / %d   %s" % (name, rows[name], hb.num_nodes, eng.solve_path(db, dm), summ["max_err_vs_f64"],
              summ["max_err_vs_f32_restatement"], summ["restatement_max_err_vs_f64"], summ["sets_differing"], summ["graphs_at_margin_risk_at_2x_error"], fam), flush=True)


if __name__ == "__main__":  # (scores_report starts worker processes: they import this module)
    main()
