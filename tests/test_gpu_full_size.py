"""GPU: the HIP kernels' output at FULL BASELINE size held against the oracle restatement DIRECTLY (no twin in between).

tests/test_full_size_parity.py does the same on the CPU with the twin's scores (all 4 000 C4 graphs); here the scores and
state bytes come from the GPU - C2, C3, one GPU's share of C4 at l = 1 and l = 20, and a C5-sized batch - and go through
oracle/parity.py: float32 and float64 evaluations of the reference's formula (gcn/layers.py:189-216,
gcn/models.py:536-573), the reference's local greedy search (heuristics.py:77-116) on the restatement's priorities
(mwis_gdpg_call.py:211-216)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# graphs solved on the GPU per configuration; from ER500 on: the any-size path at its own sizes (k_big, k_wide1, the
# layer-by-layer chain / k_big2 beyond 976 vertices)
SHARE = {"C2": 500, "C3": 500, "C4-l1": 500, "C4-l20": 500, "C5-size": 64,
         "ER500": 64, "MC900": 256, "MC900-l1": 256, "N1500": 64, "MC1500": 64}
SETS_DIFFERING = {"MC900": [182]}  # a near-tie the two float32 evaluations resolve differently; flagged by the margin test (asserted)


@pytest.mark.parametrize("name", sorted(SHARE))
def test_full_size_scores_against_the_restatement_on_gpu(engine, name):
    from distgcn_amd.engine import DeviceModel
    from oracle import parity
    _, _, model, make = parity.full_size_configs()[name]
    count = SHARE[name]
    layers = parity._model(model)
    hb = make(count, 0)
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    res = engine.solve_fused(db, dm)
    engine.check_status(res["status"])
    scores, state = res["scores"].cpu().numpy().ravel(), res["state"].cpu().numpy()
    summ, reports = parity.scores_report(name, count, hb.graph_ptr, scores, state)
    assert summ["graphs"] == count
    # (1) every score of every graph within 1e-5 of the exact (float64) evaluation, in absolute units too
    assert summ["graphs_over_1e-5_vs_f64"] == 0 and summ["abs_graphs_over_1e-5_vs_f64"] == 0, summ
    # (2) within 1e-5 of the NumPy float32 restatement too, except where that restatement is itself the one further from the
    #     exact value: the C4 share at l = 20 has one such graph (g115: 1.29e-5 from the restatement, 3.1e-6 from float64, the
    #     restatement 1.07e-5) - tests/test_full_size_parity.py lists all eight of the 4 000
    assert summ["graphs_over_1e-5_vs_f32_restatement"] == summ["of_those_restatement_further_from_f64"], summ
    assert summ["graphs_over_1e-5_vs_f32_restatement_ids"] == ([115] if name == "C4-l20" else []), summ
    # (3) the selected sets: identical to the reference's local greedy search on the restatement's priorities, every graph -
    #     except MC900's graph 182 (tests/test_full_size_parity.py), and every flip must be margin-flagged
    assert summ["sets_differing_ids"] == SETS_DIFFERING.get(name, []), summ
    assert summ["sets_differing_not_flagged_by_margin"] == 0, summ
    if name in ("ER500", "MC900", "MC900-l1", "N1500", "MC1500"):
        assert engine.solve_path(db, dm) == 2


# the other kernels and launch forms (oracle/parity.gpu_extra_configs): graphs solved on the GPU per row
EXTRA = {"C3-l3": 500, "C4-l3": 500, "C3-ld32": 128, "C3-c16-l20": 128, "BA-c64-l2": 128, "N1900": 16, "N9600-l1": 4, "N3000-l2": 8,
         "MC900-c16-l20": 64, "MC1500-c16-l4": 32, "ER600-cheb2-l2": 32, "ER200-cheb2-l1": 64}
EXTRA_PATH = {"N1900": 2, "N9600-l1": 2, "N3000-l2": 2, "MC900-c16-l20": 2, "MC1500-c16-l4": 2, "ER600-cheb2-l2": 2, "ER200-cheb2-l1": 2}


@pytest.mark.parametrize("name", sorted(EXTRA))
def test_more_shapes_against_the_restatement_on_gpu(engine, name):
    """HIP output against the restatement DIRECTLY (no twin in between) on the shapes the ten configurations above do not
    reach: the launcher's third depth, 32 input features, hidden widths 16 / 64, the fifteen-tile k_big2, k_wide1 at 9 600
    vertices and in its two-layer form, deep narrow stacks zero-padded onto k_big / k_big2, the two [I, L, L.L] checkpoints."""
    from distgcn_amd.engine import DeviceModel
    from oracle import parity
    _, _, model, make = parity.gpu_extra_configs()[name]
    count = EXTRA[name]
    layers = parity._model(model)
    hb = make(count, 0)
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    res = engine.solve_fused(db, dm)
    engine.check_status(res["status"])
    scores, state = res["scores"].cpu().numpy().ravel(), res["state"].cpu().numpy()
    summ, _ = parity.scores_report(name, count, hb.graph_ptr, scores, state)
    assert summ["graphs"] == count
    assert summ["graphs_over_1e-5_vs_f64"] == 0 and summ["abs_graphs_over_1e-5_vs_f64"] == 0, summ
    assert summ["graphs_over_1e-5_vs_f32_restatement"] == summ["of_those_restatement_further_from_f64"], summ
    assert summ["sets_differing_not_flagged_by_margin"] == 0, summ
    assert summ["sets_differing_ids"] == [], summ
    if name in EXTRA_PATH:
        assert engine.solve_path(db, dm) == EXTRA_PATH[name]
