"""TEST INFRASTRUCTURE ONLY - the envelope of float32 summation orders (oracle/f32_orders.c).

TensorFlow's float32 kernels (``gcn/layers.py:29-31, 206, 208``: ``matmul``, ``sparse_tensor_dense_matmul``, ``add_n``)
cannot be run here, so the ORDER in which they add is pinned by nothing ("parity unpinned").  This
module bounds it instead: the same formula, float32 everywhere after the feed like TF, evaluated under every order a
float32 implementation plausibly uses, next to the float64 evaluation and the kernels' own arithmetic (the twin).

``ORDERS``: name -> (entry order of a support row, spmm_fma, mm_mode)
  numpy_blas       ``oracle/ref_numpy.gcn_forward`` (SciPy csr @ dense, NumPy matmul -> BLAS sgemm): the restatement the
                   goldens hold
  coo_seq_nofma    rows in COO storage order (ascending column, the diagonal in its place: ``sparse_to_tuple`` of
                   ``sp.eye - A_hat``, ``gcn/utils.py:79-95, 258-274``), multiply and add rounded separately in both
                   products - TF's CPU ``SparseTensorDenseMatMul`` loop and a non-FMA GEMM
  coo_seq_fma      same entry order, fused multiply-add in both (an FMA build of Eigen)
  coo_fma_kblock8  / coo_fma_kblock16: the dense product's k loop split into blocks added in order (split-k GEMMs)
  coo_nofma_tree   pairwise tree over k, non-FMA
  diag_first_fma   the HIP kernels' entry order (diagonal first) with float32 FMA chains everywhere - the library's
                   arithmetic before round 3
  twin             the library's contract (include/dgcn.h "Precision"): diag_first_fma with the aggregation of layer 0 and
                   the transform of layer 1 carried in double
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from oracle import ctwin, ref_numpy as orc

ORDERS = {
    "coo_seq_nofma": ("sorted", 0, 0),
    "coo_seq_fma": ("sorted", 1, 1),
    "coo_fma_kblock8": ("sorted", 1, 2),
    "coo_fma_kblock16": ("sorted", 1, 3),
    "coo_nofma_tree": ("sorted", 0, 4),
    "diag_first_fma": ("diag_first", 1, 1),
}
ALL = ["numpy_blas"] + list(ORDERS) + ["twin"]


def _sorted_support(lap):
    """(row_ptr, col, val) with every row's entries in ascending column order (the diagonal moved to its place)."""
    rp, col, val = lap
    n = rp.size - 1
    m = sp.csr_matrix((val.copy(), col.copy(), rp.copy()), shape=(n, n))  # (sort_indices works in place)
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)


def forward_order(lap, layers, num_nodes, spmm_fma, mm_mode, X=None, x_const=None):
    L = len(layers)
    dims = np.array([layers[0]["weights"][0].shape[0]] + [l["weights"][0].shape[1] for l in layers], np.int32)
    cats = [np.ascontiguousarray(np.concatenate([np.asarray(w, np.float32) for w in l["weights"]], axis=1)) for l in layers]
    bs = [None if l.get("bias") is None else np.ascontiguousarray(l["bias"], np.float32).ravel() for l in layers]
    wptr = (C.c_void_p * L)(*[c.ctypes.data for c in cats])
    bptr = (C.c_void_p * L)(*[(b.ctypes.data if b is not None else None) for b in bs])
    acts = np.array([ctwin.ACTS[l.get("act")] for l in layers], np.int32)
    if x_const is None:
        x_const = float(np.float32(1.0 / dims[0]))
    rp, col, val = (np.ascontiguousarray(a) for a in lap)
    scores = np.empty((num_nodes, int(dims[-1])), np.float32)
    rc = ctwin.lib().ord_forward(num_nodes, ctwin._p(rp), ctwin._p(col), ctwin._p(val), L, ctwin._p(dims), wptr, bptr,
                                 ctwin._p(acts), ctwin._p(X), C.c_float(x_const), int(spmm_fma), int(mm_mode), ctwin._p(scores))
    if rc:
        raise MemoryError("ord_forward")
    return scores


def graph_envelope(indptr, indices, weights, layers, predict="mwis"):
    """One graph -> {"scores": {order: float32[n]}, "f64": float64[n], "sets": {order: bool[n]}}."""
    from distgcn_amd.batch import HostBatch
    n = int(weights.size)
    hb = HostBatch.from_csr_lists([indptr.astype(np.int64)], [indices.astype(np.int64)])
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    lap_sorted = _sorted_support(lap)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    st = orc.makestate(adj, weights.reshape(-1, 1), layers[0]["weights"][0].shape[0], 1, "gdpg", predict)
    out = {"numpy_blas": orc.gcn_forward(layers, st, np.float32)[0][:, 0]}
    f64 = orc.gcn_forward(layers, st, np.float64)[0][:, 0]
    for name, (entry, sf, mm) in ORDERS.items():
        out[name] = forward_order(lap_sorted if entry == "sorted" else lap, layers, n, sf, mm)[:, 0]
    out["twin"] = ctwin.forward(lap, layers, n)[:, 0]
    sets = {}
    for name, sc in out.items():
        state, _ = orc.lgs_vectorised(indptr, indices, orc.priority(sc, weights, predict))
        sets[name] = state == 1
    return {"scores": out, "f64": f64, "sets": sets}


def graph_envelope_summary(indptr, indices, weights, layers, predict="mwis"):
    """Per graph: each order's max |score - float64| (absolute), the largest distance between two float32 orders, the
    twin's distance from every order, and which orders select another set than the twin."""
    e = graph_envelope(indptr, indices, weights, layers, predict)
    sc, f64 = e["scores"], e["f64"]
    names = list(sc)
    err64 = {k: float(np.abs(sc[k].astype(np.float64) - f64).max()) if f64.size else 0.0 for k in names}
    f32_names = [k for k in names if k != "twin"]
    spread = 0.0
    for i, a in enumerate(f32_names):
        for b in f32_names[i + 1:]:
            spread = max(spread, float(np.abs(sc[a].astype(np.float64) - sc[b]).max()) if f64.size else 0.0)
    twin_to = {k: float(np.abs(sc["twin"].astype(np.float64) - sc[k]).max()) if f64.size else 0.0 for k in f32_names}
    differ = [k for k in f32_names if not np.array_equal(e["sets"][k], e["sets"]["twin"])]
    pr = orc.priority(sc["twin"], weights, predict)
    state = np.where(e["sets"]["twin"], 1, 2).astype(np.uint8)
    risk = int(orc.margin_risk(indptr, indices, pr, state, 2.0 * max(twin_to.values()), weights if predict == "mwis" else None))
    return {"err_vs_f64": err64, "f32_spread": spread, "twin_to": twin_to, "orders_with_another_set": differ,
            "margin_risk_at_2x_twin_distance": risk}
