"""What cannot be pinned is bounded instead: the envelope of float32 summation orders (oracle/orders.py, oracle/f32_orders.c).

The reference's float32 arithmetic runs inside TensorFlow (gcn/layers.py:29-31, 206, 208), which cannot be installed here:
"within 1e-5 of the reference's scores" can only be checked against float32 evaluations of the same formula - and those
differ among themselves.  tools/f32_envelope.py evaluated all five BASELINE configurations (9 064 graph evaluations) under seven
float32 orders + the library's contract and wrote profiles/r04_f32_order_envelope.json; here the claims drawn from it are
asserted, and re-derived on a sample that includes the worst graphs."""
import json
import os

import numpy as np
import pytest

from oracle import orders, parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_envelope_supports_the_claims():
    z = json.load(open(os.path.join(ROOT, "profiles", "r04_f32_order_envelope.json")))
    assert set(z) == {"C2", "C3", "C4-l1", "C4-l20", "C5-size"}
    assert sum(c["graphs"] for c in z.values()) == 9064
    for name, c in z.items():
        # the library's arithmetic: every score within 1e-5 (absolute) of the float64 evaluation, on every configuration
        assert c["graphs_over_1e-5_vs_f64"]["twin"] == 0 and c["max_err_vs_f64"]["twin"] < 7.4e-6, name
        # no float32 order of the envelope selects another set than the library does, on any graph
        assert c["graphs_where_some_order_selects_another_set"] == 0, name
    deep = z["C4-l20"]
    # on the hub-heavy BA graphs at l = 20 float32 evaluations of the SAME formula are further apart than 1e-5 ...
    assert deep["max_f32_spread"] > 2e-5 and deep["graphs_with_f32_spread_over_1e-5"] >= 50
    # ... and every plain float32 order that walks the rows front to back misses the exact value by more than 1e-5 somewhere
    for k in ("numpy_blas", "coo_seq_nofma", "coo_seq_fma", "diag_first_fma"):
        assert deep["graphs_over_1e-5_vs_f64"][k] >= 7 and deep["max_err_vs_f64"][k] > 1.5e-5, k
    for name in ("C2", "C3", "C4-l1", "C5-size"):  # elsewhere the envelope is narrow and the library sits inside 1e-5 of all of it
        assert z[name]["max_f32_spread"] < 1e-5 and max(z[name]["twin_max_distance_to"].values()) < 1e-5, name


@pytest.mark.parametrize("name,graphs", [("C4-l20", [115, 1670, 2280, 2770, 1320, 3945]), ("C3", [0, 7, 133]), ("C2", [5])])
def test_envelope_on_the_worst_graphs(name, graphs):
    _, _, model, make = parity.full_size_configs()[name]
    layers = parity._model(model)
    worst_spread = 0.0
    for g in graphs:
        hb = make(1, g)
        ip, ix = hb.row_ptr.astype(np.int64), hb.col_idx.astype(np.int64)
        s = orders.graph_envelope_summary(ip, ix, hb.weights, layers)
        assert s["err_vs_f64"]["twin"] <= 1e-5, (name, g, s["err_vs_f64"])
        assert s["orders_with_another_set"] == [], (name, g)
        worst_spread = max(worst_spread, s["f32_spread"])
    if name == "C4-l20":
        assert worst_spread > 1.5e-5  # g1670: 2.28e-5 between two float32 orders
