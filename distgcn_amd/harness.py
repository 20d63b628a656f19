"""The "500 graphs" evaluation loop of the reference (``mwis_dqn_test.py:304-348``) as a batched call.

The reference walks a folder of ``.mat`` files (``adj``, ``weights``, ``greedy_utility`` ...,
``Data_Generation.py:218-219``), solves each graph, and records the approximation ratio
``p = total_wt / greedy_utility`` (``mwis_dqn_test.py:321``) in a CSV.  Here the folder becomes one
block-diagonal batch and one launch; the denominators come from the stored ``greedy_utility`` or are
recomputed on the device (``greedy_search`` on raw weights, ``mwis_dqn_test.py:311``).
"""
from __future__ import annotations

import os
from struct import error as struct_error
from typing import Dict, List, Optional, Sequence

import numpy as np

from .api_common import as_csr, get_engine
from .batch import HostBatch


def load_mat_folder(path: str, limit: Optional[int] = None) -> Dict[str, list]:
    """Read the reference's dataset folders (``Data_Generation.py:218-219``: sparse ``adj``, ``weights``,
    ``greedy_utility``, ``mwis_utility``) into canonical CSR matrices.  Files are parsed by ``distgcn_amd.matfile``
    (the column-compressed arrays of a symmetric adjacency ARE its CSR arrays: no conversion, ~6x faster than
    ``scipy.io.loadmat``, which remains the fallback for anything that reader does not cover)."""
    import scipy.sparse as sp
    from . import matfile
    names = sorted(f for f in os.listdir(path) if f.endswith(".mat"))
    if limit is not None:
        names = names[:limit]
    out = {"names": names, "adjs": [], "weights": [], "greedy_utility": [], "mwis_utility": []}
    for f in names:
        full = os.path.join(path, f)
        try:
            m = matfile.loadmat(full)
            indptr, indices = matfile.symmetric_csr(m["adj"])
            n = m["adj"].shape[0]
            adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
            adj.has_sorted_indices = True
            adj.has_canonical_format = True
        except (NotImplementedError, ValueError, KeyError, struct_error):
            import scipy.io as sio
            m = sio.loadmat(full)
            adj = as_csr(m["adj"])
        out["adjs"].append(adj)
        out["weights"].append(np.asarray(m["weights"], dtype=np.float64).ravel())
        out["greedy_utility"].append(float(np.asarray(m["greedy_utility"]).ravel()[0]) if "greedy_utility" in m else None)
        out["mwis_utility"].append(float(np.asarray(m["mwis_utility"]).ravel()[0]) if "mwis_utility" in m else None)
    return out


def greedy_utilities(adjs: Sequence, wts_list: Sequence) -> np.ndarray:
    """``greedy_search(adj, wts)`` totals for a batch (the denominators of ``mwis_dqn_test.py:311``)."""
    import torch
    eng = get_engine()
    csrs = [as_csr(a) for a in adjs]
    hb = HostBatch.from_csr_lists([c.indptr for c in csrs], [c.indices for c in csrs],
                                  [np.asarray(w, dtype=np.float64).ravel() for w in wts_list])
    db = eng.upload(hb)
    res = eng.lgs(db, prio=db.weights, sum_weights=db.weights)
    eng.check_status(res["status"])
    return res["totals"].cpu().numpy()


def _solve_with_exploration(agent, adjs, wts_list, epsilon: float, rng):
    """The ``test=False`` branch of the reference's loop (``mwis_dqn_test.py:244-256``): with probability ``epsilon`` a
    graph's GCN scores are replaced by ``uniform(0, 1)`` draws before the priority product and the greedy search.
    ``rng`` supplies ``rand()`` / ``uniform(size=)`` (``numpy.random`` itself, seeded by the caller, replays the
    reference's draws graph by graph: one ``rand()``, then ``uniform(size=n)`` when it fires).  Batched: one forward
    pass and one greedy launch; only the score rows of the graphs that fire are rewritten on the host."""
    import torch
    eng = get_engine()
    csrs = [as_csr(a) for a in adjs]
    wl = [np.asarray(w, dtype=np.float64).ravel() for w in wts_list]
    hb = HostBatch.from_csr_lists([c.indptr for c in csrs], [c.indices for c in csrs], wl)
    db = eng.upload(hb)
    dm = agent.model.device_model(eng)
    # mwis_dqn_test.py:162-169: features ones * w row-normalised -> 1/F on positive-weight rows, 0 on zero-weight rows
    F = agent.model.input_dim
    X = None
    if hb.num_nodes and np.any(hb.weights <= 0):
        X = torch.from_numpy(np.where(hb.weights[:, None] > 0, np.float32(1.0 / F), np.float32(0.0)).astype(np.float32)
                             .repeat(F, axis=1)).to(eng.device)
    mode = 1 if eng.solve_supported(db, dm) else 0
    scores = agent.model.forward_batch(eng, db, X=X, mode=mode)
    host = None
    for n0, n1 in hb.graph_slices():
        if rng.rand() <= epsilon:
            if host is None:
                host = scores.cpu().numpy().copy()
            host[n0:n1, 0] = rng.uniform(size=n1 - n0)
    if host is not None:
        scores = torch.from_numpy(host).to(eng.device)
    predict = getattr(agent.flags, "predict", "mwis")
    res = eng.lgs(db, scores=scores, weights=db.weights if predict == "mwis" else None, sum_weights=db.weights)
    eng.check_status(res["status"])
    st, tot = res["state"].cpu().numpy(), res["totals"].cpu().numpy()
    return [(set(int(i) for i in np.flatnonzero(st[n0:n1] == 1)), np.float64(tot[g]))
            for g, (n0, n1) in enumerate(hb.graph_slices())]


def evaluate(agent, adjs: Sequence, wts_list: Sequence, greedy_utility: Optional[Sequence[float]] = None,
             names: Optional[Sequence[str]] = None, epsilon: float = 0.0, rng=None) -> List[dict]:
    """Solve every graph with ``agent.solve_mwis_batch`` and return one record per graph:
    ``{"data", "p", "total", "size"}`` - ``p`` is the reference's ratio column.  ``epsilon > 0`` takes the
    reference's exploring branch (the scripts run ``solve_mwis(test=False)`` with ``--epsilon=.0002``)."""
    if epsilon > 0.0:
        res = _solve_with_exploration(agent, adjs, wts_list, epsilon, rng if rng is not None else np.random)
    else:
        res = agent.solve_mwis_batch(adjs, wts_list)
    if greedy_utility is None or any(g is None for g in greedy_utility):
        greedy_utility = greedy_utilities(adjs, wts_list)
    rows = []
    for i, r in enumerate(res):
        total = float(r[1])
        rows.append({"data": names[i] if names else i, "p": total / float(greedy_utility[i]), "total": total,
                     "size": len(r[0])})
    return rows


def write_csv(rows: List[dict], path: str) -> None:
    """``results.to_csv('./output/<model>.csv')`` (``mwis_dqn_test.py:348``): columns data, p."""
    import csv
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["", "data", "p"])
        for i, r in enumerate(rows):
            w.writerow([i, r["data"], r["p"]])
