"""Drop-in for the greedy functions of the reference's ``heuristics.py``, on the GPU.

Same names, arguments and return types (``set`` of ints, ``numpy.float64`` totals).  Every variant
maps onto one kernel, ``dgcn_lgs_batch`` (one workgroup per graph); a single graph is a batch of
one.  The LP/MIP baselines of the reference (``heuristics.py:308-484``: Gurobi, GLPK, igraph) are
out of scope.

Preconditions as in the reference: no self-loops; priorities must not be NaN (the reference never
terminates on either, ``heuristics.py:94,103-111``; here both raise ``DgcnError``).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .api_common import get_engine, single_batch


_native = {}


def _run_native(eng, adj, w):
    """The plain search (all rounds, no statistics) behind two native calls: ``dgcn_host_solver_*`` with no model."""
    from .api_common import as_csr
    hs = _native.get(id(eng))
    if hs is None:
        from .serving import HostSolver
        hs = _native[id(eng)] = HostSolver(eng, None, depth=1)
    a = as_csr(adj)
    if a.shape[0] != w.size:
        raise ValueError("adjacency has %d vertices, weights %d" % (a.shape[0], w.size))
    with hs.lock:  # (another thread's call would overwrite the slot's result buffer)
        h = hs.solve([a.indptr], [a.indices], [w])
        state = h["state"].copy()
        return {"state": state, "mwis": set(np.flatnonzero(state == 1).tolist()), "total": np.float64(h["totals"][0]),
                "rounds": int(h["rounds"][0])}


def _run(adj, wts, max_rounds=0, want_stats=False, want_overhead=False):
    eng = get_engine()
    w = np.array(wts, dtype=np.float64).flatten()
    if max_rounds == 0 and not want_stats and not want_overhead and w.size:
        try:
            return _run_native(eng, adj, w)
        except (TypeError, BufferError):  # index arrays the native packer does not take: the NumPy path below
            pass
    hb = single_batch(adj)
    if hb.num_nodes != w.size:
        raise ValueError("adjacency has %d vertices, weights %d" % (hb.num_nodes, w.size))
    if w.size == 0:  # the reference's loops simply do not run
        out = {"state": np.zeros(0, np.uint8), "mwis": set(), "total": np.float64(0.0), "rounds": 0, "p2p": 0, "bst": 0}
        out["overhead"] = np.zeros(0)
        return out
    if hb.num_nodes and np.any(hb.col_idx == np.repeat(np.arange(hb.num_nodes), np.diff(hb.row_ptr))):
        raise _lib.DgcnError("adjacency has a self-loop (heuristics.py:94 would never terminate)")
    hb.weights = w  # the priorities ride along in the batch's single host-to-device copy
    db = eng.upload(hb)
    res = eng.lgs(db, prio=db.weights, max_rounds=max_rounds, want_stats=want_stats, want_overhead=want_overhead)
    h = eng.fetch_packed(res)  # one device-to-host copy
    eng.check_status_bits(int(h["status"][0]))
    state = h["state"][:w.size]
    out = {"state": state, "mwis": set(int(i) for i in np.flatnonzero(state == 1)),
           "total": np.float64(h["totals"][0]), "rounds": int(h["rounds"][0])}
    if want_stats or want_overhead:
        out["p2p"], out["bst"] = int(h["stats"][0]), int(h["stats"][1])
    if want_overhead:
        out["overhead"] = h["overhead"][:w.size].astype(np.float64)
    return out


def local_greedy_search(adj, wts):
    """``heuristics.py:77-116`` -> (mwis, total_ws)."""
    r = _run(adj, wts)
    return r["mwis"], r["total"]


def local_greedy_search_count(adj, wts):
    """``heuristics.py:119-160`` -> (mwis, total_ws, step)."""
    r = _run(adj, wts)
    return r["mwis"], r["total"], r["rounds"]


def local_greedy_search_stats(adj, wts):
    """``heuristics.py:163-209`` -> (mwis, total_ws, step, p2p, bst)."""
    r = _run(adj, wts, want_stats=True)
    return r["mwis"], r["total"], r["rounds"], r["p2p"], r["bst"]


def local_greedy_search_overhead(adj, wts):
    """``heuristics.py:212-263`` -> (mwis, total_ws, step, p2p, bst, oh_vec)."""
    r = _run(adj, wts, want_stats=True, want_overhead=True)
    return r["mwis"], r["total"], r["rounds"], r["p2p"], r["bst"], r["overhead"]


def local_greedy_search_nstep(adj, wts, nstep=1):
    """``heuristics.py:266-305`` -> (mwis, total_ws, nb_is) after at most ``nstep`` rounds.
    ``nstep=0`` runs no round (the reference's ``while ... and step`` loop)."""
    if nstep == 0:
        return set(), np.float64(0.0), set()
    r = _run(adj, wts, max_rounds=int(nstep))
    return r["mwis"], r["total"], set(int(i) for i in np.flatnonzero(r["state"] == 2))


def greedy_search(adj, wts):
    """``heuristics.py:13-35``: sort by weight, sweep.  The sweep selects the lexicographically first
    maximal independent set under (weight desc, index asc) - exactly what the round-synchronous local
    search converges to, so it runs on the same kernel.  (The reference's unstable ``argsort`` leaves
    the order of EQUAL weights unspecified; ties are broken by index here.)"""
    r = _run(adj, wts)
    return r["mwis"], r["total"]
