"""Full-size parity report: a score vector (HIP or twin) against the oracle restatement, graph by graph.
TEST / BENCH INFRASTRUCTURE ONLY - the checker, never the thing measured or shipped.

For every graph of a batch: max |score - float32 restatement|, max |score - float64 restatement| and the float32
restatement's own distance from float64 (oracle/ref_numpy.gcn_forward on the reference's makestate, one graph per call
like the reference; errors both in units of max(1, |score|), see tests/conftest.check_scores, and in absolute score units); the set the reference's local
greedy search picks on the RESTATEMENT's priorities (ref_numpy.lgs_vectorised, pinned against the imported
heuristics.local_greedy_search) against the given states; and SURVEY 7.3(c)'s margin count at delta = twice the
measured error.  `full_size_configs()` names the BASELINE configurations (C2, C3, C4 at l=1 and l=20, a C5-sized batch).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_numpy as orc  # noqa: E402


def graph_report(indptr, indices, weights, layers, scores, state=None, predict="mwis", flavour="gdpg"):
    """One graph -> dict(e32, e64, e3264, set_differs, risk_at_2e).  `state`: membership bytes (1 = member) to compare
    with the reference search on the restatement's priorities; None: the search is run on `scores` here."""
    n = int(weights.size)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    st = orc.makestate(adj, weights.reshape(-1, 1), layers[0]["weights"][0].shape[0], len(layers[0]["weights"]) - 1, flavour, predict)
    f32 = orc.gcn_forward(layers, st, np.float32)[0][:, 0]
    f64 = orc.gcn_forward(layers, st, np.float64)[0][:, 0]
    scores = np.asarray(scores, dtype=np.float32).ravel()
    unit = np.maximum(1.0, np.abs(f64))
    e32 = float((np.abs(scores.astype(np.float64) - f32) / unit).max()) if n else 0.0
    e64 = float((np.abs(scores - f64) / unit).max()) if n else 0.0
    e3264 = float((np.abs(f32.astype(np.float64) - f64) / unit).max()) if n else 0.0
    # the same three in ABSOLUTE score units (they differ from the scaled ones only where |score| > 1: BA hubs reach 2.2)
    a32 = float(np.abs(scores.astype(np.float64) - f32).max()) if n else 0.0
    a64 = float(np.abs(scores - f64).max()) if n else 0.0
    a3264 = float(np.abs(f32.astype(np.float64) - f64).max()) if n else 0.0
    pr_ref = orc.priority(f32, weights, predict)
    ref_state, _ = orc.lgs_vectorised(indptr, indices, pr_ref)
    pr_got = orc.priority(scores, weights, predict)
    if state is None:
        state, _ = orc.lgs_vectorised(indptr, indices, pr_got)
    differs = not np.array_equal(np.asarray(state) == 1, ref_state == 1)
    wabs = weights if predict == "mwis" else None
    risk = orc.margin_risk(indptr, indices, pr_got, np.asarray(state), 2.0 * e32, wabs)
    return {"e32": e32, "e64": e64, "e3264": e3264, "a32": a32, "a64": a64, "a3264": a3264, "set_differs": bool(differs),
            "risk_at_2e": int(risk)}


def batch_report(hb, layers, scores, state=None, predict="mwis", flavour="gdpg", graphs=None):
    """Per-graph reports of a HostBatch (all graphs, or the listed ones)."""
    out = []
    slices = hb.graph_slices()
    for g in (range(hb.num_graphs) if graphs is None else graphs):
        n0, n1 = slices[g]
        ip = hb.row_ptr[n0:n1 + 1].astype(np.int64) - int(hb.row_ptr[n0])
        ix = hb.col_idx[hb.row_ptr[n0]:hb.row_ptr[n1]].astype(np.int64) - n0
        out.append(graph_report(ip, ix, hb.weights[n0:n1], layers, scores[n0:n1], None if state is None else state[n0:n1],
                                predict, flavour))
    return out


def summarize(reports):
    e32 = np.array([r["e32"] for r in reports])
    e64 = np.array([r["e64"] for r in reports])
    e3264 = np.array([r["e3264"] for r in reports])
    over = e32 > 1e-5
    return {"graphs": len(reports),
            "max_err_vs_f32_restatement": float(e32.max()), "graphs_over_1e-5_vs_f32_restatement": int(over.sum()),
            "of_those_restatement_further_from_f64": int((over & (e3264 > e64)).sum()),
            "max_err_vs_f64": float(e64.max()), "graphs_over_1e-5_vs_f64": int((e64 > 1e-5).sum()),
            "restatement_max_err_vs_f64": float(e3264.max()),
            "graphs_over_1e-5_vs_f32_restatement_ids": [int(i) for i in np.flatnonzero(over)],
            # absolute score units (the scaled figures above divide by max(1, |score|))
            "abs_max_err_vs_f32_restatement": float(max(r["a32"] for r in reports)),
            "abs_graphs_over_1e-5_vs_f32_restatement": int(sum(r["a32"] > 1e-5 for r in reports)),
            "abs_graphs_over_1e-5_vs_f32_restatement_ids": [i for i, r in enumerate(reports) if r["a32"] > 1e-5],
            "abs_max_err_vs_f64": float(max(r["a64"] for r in reports)),
            "abs_graphs_over_1e-5_vs_f64": int(sum(r["a64"] > 1e-5 for r in reports)),
            "abs_restatement_max_err_vs_f64": float(max(r["a3264"] for r in reports)),
            "sets_differing": int(sum(r["set_differs"] for r in reports)),
            "sets_differing_not_flagged_by_margin": int(sum(r["set_differs"] and r["risk_at_2e"] == 0 for r in reports)),
            "sets_differing_ids": [i for i, r in enumerate(reports) if r["set_differs"]],
            "graphs_at_margin_risk_at_2x_error": int(sum(r["risk_at_2e"] > 0 for r in reports))}


# ---------------------------------------------------------------------------------------------- the BASELINE configurations
def _model(name):
    from distgcn_amd.gcn.models import layers_from_params
    z = np.load(os.path.join(ROOT, "tests", "golden", "all_models.npz"))
    pre = name + "|"
    return layers_from_params({k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and "graphconvolution" in k})


def full_size_configs():
    """name -> (family, graphs, model fixture name, batch maker(count, first))."""
    from distgcn_amd import datagen
    m = "result_%s_deep_ld1_c32_l%d_cheb1_diver1_mwis_dqn"
    return {
        "C2": ("500 ER N=100 p=0.1, IS4SAT l=1", 500, m % ("IS4SAT", 1), lambda c, f: datagen.er_batch(c, 100, 0.1, first_index=f)),
        "C3": ("500 ER N=200 p=0.1, IS4SAT l=20", 500, m % ("IS4SAT", 20), lambda c, f: datagen.er_batch(c, 200, 0.1, first_index=f)),
        "C4-l1": ("4000 BA test2 mix, DQNBA l=1", 4000, m % ("DQNBA", 1), lambda c, f: datagen.ba_test2_batch(c, first_index=f)),
        "C4-l20": ("4000 BA test2 mix, DQNBA l=20", 4000, m % ("DQNBA", 20), lambda c, f: datagen.ba_test2_batch(c, first_index=f)),
        "C5-size": ("64 ER N=500 p=0.02, IS4SAT l=20", 64, m % ("IS4SAT", 20), lambda c, f: datagen.er_batch(c, 500, 0.02, first_index=f)),
        # the any-size path's own sizes (csrc/big.hip, wide.hip, the layer-by-layer chain beyond 976 vertices): the shapes of
        # bench.py --config ER500 / MC900 / MC900-l1 / MC1500 and a sparse 1 500-vertex ER batch
        "ER500": ("64 ER N=500 p=0.1, IS4SAT l=20", 64, m % ("IS4SAT", 20), lambda c, f: datagen.er_batch(c, 500, 0.1, first_index=f)),
        "MC900": ("256 joint 3 x 300-flow conflict graphs, IS4SAT l=20", 256, m % ("IS4SAT", 20),
                  lambda c, f: datagen.multichannel_batch(c, 300, 0.03, first_index=f)),
        "MC900-l1": ("256 joint 3 x 300-flow conflict graphs, IS4SAT l=1", 256, m % ("IS4SAT", 1),
                     lambda c, f: datagen.multichannel_batch(c, 300, 0.03, first_index=f)),
        "N1500": ("64 ER N=1500 p=0.004, IS4SAT l=20", 64, m % ("IS4SAT", 20), lambda c, f: datagen.er_batch(c, 1500, 0.004, first_index=f)),
        "MC1500": ("64 joint 3 x 500-flow conflict graphs, IS4SAT l=20", 64, m % ("IS4SAT", 20),
                   lambda c, f: datagen.multichannel_batch(c, 500, 0.03, first_index=f)),
    }


def gpu_extra_configs():
    """More (family, graphs, model fixture, batch maker) rows, held against the restatement by the -m gpu suite only
    (tests/test_gpu_full_size.py) - the CPU suite's twin runs stay at full_size_configs().  The shapes the other kernels
    and launch forms take: the launcher's third depth (bash/generalization_dqn_test.sh:20-33: layers 1, 3, 20), 32 input
    features, hidden widths 16 and 64, the fifteen-tile k_big2, k_wide1 at its largest sizes (one and two layers), deep
    stacks narrower than 32 on k_big / k_big2, and the two [I, L, L.L] checkpoints."""
    from distgcn_amd import datagen
    m = "result_%s_deep_ld%d_c%d_l%d_cheb%d_diver1_mwis_dqn"
    er = lambda n, p: (lambda c, f: datagen.er_batch(c, n, p, first_index=f))
    ba = lambda c, f: datagen.ba_test2_batch(c, first_index=f)
    mc = lambda nf: (lambda c, f: datagen.multichannel_batch(c, nf, 0.03, first_index=f))
    return {
        "C3-l3": ("ER N=200 p=0.1, IS4SAT l=3", 500, m % ("IS4SAT", 1, 32, 3, 1), er(200, 0.1)),
        "C4-l3": ("BA test2 mix, DQNBA l=3", 500, m % ("DQNBA", 1, 32, 3, 1), ba),
        "C3-ld32": ("ER N=200 p=0.1, IS4SAT 32 input features l=20", 128, m % ("IS4SAT", 32, 32, 20, 1), er(200, 0.1)),
        "C3-c16-l20": ("ER N=200 p=0.1, IS4SAT c=16 l=20", 128, m % ("IS4SAT", 1, 16, 20, 1), er(200, 0.1)),
        "BA-c64-l2": ("BA test2 mix, IS4SAT c=64 l=2", 128, m % ("IS4SAT", 1, 64, 2, 1), ba),
        "N1900": ("ER N=1900 p=0.004, IS4SAT l=20 (k_big2, fifteen tiles)", 16, m % ("IS4SAT", 1, 32, 20, 1), er(1900, 0.004)),
        "N9600-l1": ("ER N=9600 p=0.0005, IS4SAT l=1 (k_wide1)", 4, m % ("IS4SAT", 1, 32, 1, 1), er(9600, 0.0005)),
        "N3000-l2": ("ER N=3000 p=0.002, IS4SAT l=2 (k_wide1, two layers)", 8, m % ("IS4SAT", 1, 32, 2, 1), er(3000, 0.002)),
        "MC900-c16-l20": ("joint 3 x 300-flow conflict graphs, IS4SAT c=16 l=20 (zero-padded onto k_big)", 64, m % ("IS4SAT", 1, 16, 20, 1), mc(300)),
        "MC1500-c16-l4": ("joint 3 x 500-flow conflict graphs, IS4SAT c=16 l=4 (zero-padded onto k_big2)", 32, m % ("IS4SAT", 1, 16, 4, 1), mc(500)),
        "ER600-cheb2-l2": ("ER N=600 p=0.02, IS4SAT c=1 l=2 [I, L, L.L]", 32, m % ("IS4SAT", 1, 1, 2, 2), er(600, 0.02)),
        "ER200-cheb2-l1": ("ER N=200 p=0.1, IS4SAT c=1 l=1 [I, L, L.L]", 64, m % ("IS4SAT", 1, 1, 1, 2), er(200, 0.1)),
    }


def all_configs():
    return {**full_size_configs(), **gpu_extra_configs()}


def _chunk(args):
    """Worker: graphs [first, first + count) of a configuration through the C twin (bit-equal to the HIP kernels: the
    -m gpu suite proves it on these very batches) and the report above."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    name, first, count = args
    from oracle import ctwin
    _, _, model, make = full_size_configs()[name]
    layers = _model(model)
    hb = make(count, first)
    res = ctwin.solve(hb, layers)
    return batch_report(hb, layers, res["scores"][:, 0], res["state"])


def _chunk_given(args):
    """Worker: graphs [first, first + count) of a configuration, scores / states GIVEN (fetched from the GPU by the caller)."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    name, first, count, scores, state = args
    _, _, model, make = all_configs()[name]
    return batch_report(make(count, first), _model(model), scores, state)


def scores_report(name, count, graph_ptr, scores, state, procs=None, chunk=64):
    """The first ``count`` graphs of a configuration with the scores / state bytes the caller computed for that very batch
    (``full_size_configs()[name][3](count, 0)``: the generators are seeded per graph) -> (summary dict, per-graph reports).
    This is how the -m gpu suite holds the HIP kernels' output against the restatement directly."""
    graph_ptr = np.asarray(graph_ptr)
    scores, state = np.asarray(scores, np.float32).ravel(), np.asarray(state)
    jobs = []
    for f in range(0, count, chunk):
        c = min(chunk, count - f)
        n0, n1 = int(graph_ptr[f]), int(graph_ptr[f + c])
        jobs.append((name, f, c, scores[n0:n1].copy(), state[n0:n1].copy()))
    procs = procs or min(len(jobs), max(1, os.cpu_count() or 2), 16)
    if procs <= 1:
        parts = [_chunk_given(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(procs) as pool:
            parts = pool.map(_chunk_given, jobs)
    reports = [r for p in parts for r in p]
    return summarize(reports), reports


def twin_report(name, procs=None, chunk=125):
    """The whole configuration on the CPU twin, in forked workers -> (summary dict, per-graph reports)."""
    _, total, _, _ = full_size_configs()[name]
    chunk = min(chunk, max(8, (total + 7) // 8))  # (small configurations still use every worker)
    jobs = [(name, f, min(chunk, total - f)) for f in range(0, total, chunk)]
    procs = procs or min(len(jobs), max(1, (os.cpu_count() or 2) - 0), 8)
    if procs <= 1:
        parts = [_chunk(j) for j in jobs]
    else:
        # fresh interpreters, not forks: the caller (a pytest process that has run BLAS calls, gloo ranks and the native
        # packer's worker pool) holds threads and locks a forked child would inherit half of
        with mp.get_context("spawn").Pool(procs) as pool:
            parts = pool.map(_chunk, jobs)
    reports = [r for p in parts for r in p]
    return summarize(reports), reports
