"""Parity at FULL BASELINE sizes against the oracle restatement (not the twin): CPU side.

The -m gpu suite proves the HIP kernels bit-equal to the C twin on these very batches (test_solve_full_size_vs_twin,
test_c4_full_batch_and_its_eight_shards) and holds their output against the restatement directly
(tests/test_gpu_full_size.py::test_full_size_scores_against_the_restatement_on_gpu); here the twin's
scores and sets for EVERY graph of C2, C3, C4 (all 4 000 graphs, DQNBA l=1 and l=20) and a C5-sized batch are held
against oracle/ref_numpy (float32 and float64 restatements of the reference's formula, one graph per call) and the
reference's local greedy search on the restatement's priorities - and, since round 5, the any-size path's own sizes: ER500,
MC900 (l = 20 and the multi-channel launcher's l = 1), a sparse 1 500-vertex ER batch and the joint 3 x 500-flow graphs.  oracle/parity.py does the work in forked
processes (about a minute on 8 cores); tools/parity_full_size.py writes the same numbers to profiles/.
"""
import numpy as np
import pytest

from oracle import parity

CONFIGS = list(parity.full_size_configs())
# C4 at l = 20: the graphs whose twin (= HIP) scores are more than 1e-5 from the NumPy float32 restatement - on every one of
# them that restatement is the one further from the float64 evaluation (asserted below)
C4_L20_OVER_F32 = [115, 945, 995, 1105, 1670, 1945, 2280, 2770]
C4_L20_OVER_F32_ABS = [115, 890, 945, 995, 1105, 1670, 1945, 2280, 2770]
SETS_DIFFERING = {"MC900": [182]}  # graphs whose set differs from the search on the float32 restatement's priorities (all margin-flagged)


@pytest.mark.parametrize("name", CONFIGS)
def test_full_size_scores_and_sets_against_the_restatement(name):
    summ, reports = parity.twin_report(name)
    assert summ["graphs"] == parity.full_size_configs()[name][1]
    # (1) every score of every graph within 1e-5 of the exact (float64) evaluation - strict
    assert summ["graphs_over_1e-5_vs_f64"] == 0, summ
    # (2) within 1e-5 of the float32 restatement too, except on graphs where that restatement is itself the one
    #     further from the exact value (C4, l = 20: a handful of hub-heavy BA graphs; its own error reaches 1.8e-5)
    assert summ["graphs_over_1e-5_vs_f32_restatement"] == summ["of_those_restatement_further_from_f64"], summ
    if name != "C4-l20":
        assert summ["graphs_over_1e-5_vs_f32_restatement"] == 0 and summ["abs_graphs_over_1e-5_vs_f32_restatement"] == 0, summ
    else:
        # the measured figures, not an allowance: eight graphs in units of max(1, |score|), nine in absolute units (g890 has
        # a score above 1), the largest difference 1.609e-5 on g1670 (BA N = 300, m = 2) - where the float32 restatement is
        # 1.780e-5 from the exact value and the kernels 5.0e-6.  profiles/r04_f32_order_envelope.json: seven float32
        # summation orders of the same formula differ from ONE ANOTHER by up to 2.28e-5 on this batch (50 graphs > 1e-5).
        assert summ["graphs_over_1e-5_vs_f32_restatement_ids"] == C4_L20_OVER_F32, summ
        assert summ["abs_graphs_over_1e-5_vs_f32_restatement_ids"] == C4_L20_OVER_F32_ABS, summ
        assert summ["max_err_vs_f32_restatement"] <= 1.61e-5 and summ["abs_max_err_vs_f32_restatement"] <= 1.61e-5, summ
        assert summ["abs_max_err_vs_f64"] <= 7.4e-6 and summ["abs_graphs_over_1e-5_vs_f64"] == 0, summ
        assert summ["max_err_vs_f64"] < 0.5 * summ["restatement_max_err_vs_f64"], summ  # the kernels' order is the more exact one
    # (3) selected sets: identical to the reference's local_greedy_search on the restatement's priorities, all graphs -
    #     but ONE: graph 182 of the 256 joint 3 x 300-flow graphs (MC900, l = 20) holds a near-tie the two float32 evaluations
    #     resolve differently (scores within 8.3e-6 of each other; SURVEY 7.3 predicted such flips at this scale).  Every flip
    #     must be one the margin test flags (an excluded vertex whose exclusion does not survive twice the measured error) -
    #     a flip it does not flag would be a wrong set, not a near-tie
    assert summ["sets_differing_ids"] == SETS_DIFFERING.get(name, []), summ
    assert summ["sets_differing_not_flagged_by_margin"] == 0, summ
    # the per-graph margin report is consistent: a graph whose set could change under twice the measured error is flagged
    assert all(r["risk_at_2e"] >= 0 for r in reports)
