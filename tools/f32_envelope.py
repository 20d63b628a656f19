#!/usr/bin/env python3
"""The envelope of float32 summation orders on every BASELINE configuration (oracle/orders.py, oracle/f32_orders.c).

TensorFlow's float32 kernels cannot be executed here, so their summation order is unpinned; this bounds it.  For every
graph of C2, C3, C4 (l=1, l=20; all 4 000 graphs) and the C5-sized batch: the reference's formula in float32 under eight
orders (NumPy/BLAS; COO storage order without / with FMA; split-k by 8 / 16; pairwise tree; the kernels' entry order in
float32; the library's contract = the twin), each one's distance from the float64 evaluation, the largest distance
between two float32 orders, the twin's distance from each, and the selected sets under each order's priorities.

    python tools/f32_envelope.py [--configs C3,C4-l20] [--limit N] [--out profiles/r04_f32_order_envelope.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _chunk(args):
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    name, first, count = args
    from oracle import orders, parity
    _, _, model, make = parity.full_size_configs()[name]
    layers = parity._model(model)
    hb = make(count, first)
    out = []
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        ip = hb.row_ptr[n0:n1 + 1].astype(np.int64) - int(hb.row_ptr[n0])
        ix = hb.col_idx[hb.row_ptr[n0]:hb.row_ptr[n1]].astype(np.int64) - n0
        s = orders.graph_envelope_summary(ip, ix, hb.weights[n0:n1], layers)
        s["graph"] = first + g
        s["n"] = int(n1 - n0)
        out.append(s)
    return out


def summarize(name, reps):
    from oracle import orders
    names = orders.ALL
    f32 = [k for k in names if k != "twin"]
    res = {"graphs": len(reps), "units": "absolute score units"}
    res["max_err_vs_f64"] = {k: max(r["err_vs_f64"][k] for r in reps) for k in names}
    res["graphs_over_1e-5_vs_f64"] = {k: sum(r["err_vs_f64"][k] > 1e-5 for r in reps) for k in names}
    res["max_f32_spread"] = max(r["f32_spread"] for r in reps)
    res["graphs_with_f32_spread_over_1e-5"] = sum(r["f32_spread"] > 1e-5 for r in reps)
    res["twin_max_distance_to"] = {k: max(r["twin_to"][k] for r in reps) for k in f32}
    res["twin_graphs_over_1e-5_from"] = {k: sum(r["twin_to"][k] > 1e-5 for r in reps) for k in f32}
    # is the twin ever further from an order than the orders are from one another?
    res["graphs_where_twin_is_outside_the_f32_spread"] = sum(max(r["twin_to"].values()) > r["f32_spread"] for r in reps)
    diff = [r for r in reps if r["orders_with_another_set"]]
    res["graphs_where_some_order_selects_another_set"] = len(diff)
    res["of_those_flagged_by_margin_risk"] = sum(r["margin_risk_at_2x_twin_distance"] > 0 for r in diff)
    res["those_graphs"] = [{"graph": r["graph"], "n": r["n"], "orders": r["orders_with_another_set"],
                            "margin_risk": r["margin_risk_at_2x_twin_distance"]} for r in diff][:50]
    over = sorted((r for r in reps if r["f32_spread"] > 1e-5), key=lambda r: -r["f32_spread"])[:12]
    res["largest_spreads"] = [{"graph": r["graph"], "n": r["n"], "f32_spread": r["f32_spread"],
                               "twin_vs_f64": r["err_vs_f64"]["twin"], "numpy_blas_vs_f64": r["err_vs_f64"]["numpy_blas"]} for r in over]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C4-l1,C4-l20,C5-size")
    ap.add_argument("--limit", type=int, default=0, help="graphs per configuration (0 = all)")
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_f32_order_envelope.json"))
    a = ap.parse_args()
    from oracle import parity
    out = {}
    for name in a.configs.split(","):
        what, total, _, _ = parity.full_size_configs()[name]
        if a.limit:
            total = min(total, a.limit)
        jobs = [(name, f, min(50, total - f)) for f in range(0, total, 50)]
        with mp.get_context("spawn").Pool(a.procs) as pool:
            parts = pool.map(_chunk, jobs)
        reps = [r for p in parts for r in p]
        out[name] = summarize(name, reps)
        out[name]["workload"] = what
        print(name, json.dumps({k: v for k, v in out[name].items() if k not in ("those_graphs", "largest_spreads")}), flush=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
