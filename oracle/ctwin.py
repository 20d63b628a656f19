"""ctypes wrapper of oracle/dgcn_oracle.c (the CPU twin).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg - never by
the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("dgcn_oracle.c", "f32_orders.c", "Makefile")]
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", "_build/liboracle.so"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()  # (re)built when a source is newer than the library
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dinv_table(max_degree):
    d = np.arange(max_degree + 1, dtype=np.float64)
    with np.errstate(divide="ignore"):
        t = np.power(d, -0.5)
    t[np.isinf(t)] = 0.0
    return t


def supports(graph_ptr, row_ptr, col_idx):
    graph_ptr = np.ascontiguousarray(graph_ptr, np.int32)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    n = int(graph_ptr[-1])
    e = int(row_ptr[-1])
    deg = np.diff(row_ptr)
    tab = dinv_table(int(deg.max()) if deg.size else 0)
    lrp = np.empty(n + 1, np.int32)
    lc = np.empty(n + e, np.int32)
    lv = np.empty(n + e, np.float32)
    fault = lib().orc_supports(n, _p(graph_ptr), int(graph_ptr.size - 1), _p(row_ptr), _p(col_idx), _p(tab),
                               int(tab.size), _p(lrp), _p(lc), _p(lv))
    return lrp, lc, lv, fault


def supports2(graph_ptr, row_ptr, col_idx):
    """T_2 = L.L (explicit, SciPy csr_matmat order; see dgcn_oracle.c) -> (row_ptr2, col2, val2 float32, fault)."""
    graph_ptr = np.ascontiguousarray(graph_ptr, np.int32)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    n = int(graph_ptr[-1])
    deg = np.diff(row_ptr)
    tab = dinv_table(int(deg.max()) if deg.size else 0)
    rp2 = np.zeros(n + 1, np.int32)
    args = (n, _p(graph_ptr), int(graph_ptr.size - 1), _p(row_ptr), _p(col_idx), _p(tab), int(tab.size), _p(rp2))
    lib().orc_supports2(*args, None, None)
    nnz = int(rp2[n])
    c2 = np.empty(max(nnz, 1), np.int32)
    v2 = np.empty(max(nnz, 1), np.float32)
    fault = lib().orc_supports2(*args, _p(c2), _p(v2))
    return rp2, c2[:nnz], v2[:nnz], fault


def spmm_split(C_feat):
    return int(lib().orc_spmm_split(int(C_feat)))


def spmm(row_ptr, col_idx, values, Z, C_feat, Y0=None, bias=None, act=0, split=0, precise=False):
    """``precise``: the row sum as one fma chain in double (the contract of layer index 0, DGCN_SPMM_PRECISE)."""
    n = row_ptr.size - 1
    Z = np.ascontiguousarray(Z, np.float32)
    Y = np.empty((n, C_feat), np.float32)
    if Y0 is not None:
        Y0 = np.ascontiguousarray(Y0, np.float32)
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
    if precise:
        lib().orc_spmm_f64(n, _p(row_ptr), _p(col_idx), _p(values), _p(Z), int(Z.shape[1]), C_feat, _p(Y0),
                           int(Y0.shape[1]) if Y0 is not None else 0, _p(bias), act, _p(Y), C_feat)
        return Y
    lib().orc_spmm(n, _p(row_ptr), _p(col_idx), _p(values), _p(Z), int(Z.shape[1]), C_feat, _p(Y0),
                   int(Y0.shape[1]) if Y0 is not None else 0, _p(bias), act, _p(Y), C_feat, int(split))
    return Y


def transform(H, W, rows=None, h_const=1.0, precise=False):
    """``precise``: the k chain carried in double, rounded once (the contract of layer index 1)."""
    W = np.ascontiguousarray(W, np.float32)
    cin, ctot = W.shape
    if H is not None:
        H = np.ascontiguousarray(H, np.float32)
        rows = H.shape[0]
    Z = np.empty((rows, ctot), np.float32)
    fn = lib().orc_transform_f64 if precise else lib().orc_transform
    fn(_p(H), int(H.shape[1]) if H is not None else cin, C.c_float(h_const), rows, cin, _p(W), ctot, _p(Z), ctot)
    return Z


ACTS = {"linear": 0, "identity": 0, None: 0, "leaky_relu": 1, "relu": 2}


def forward(lap, layers, num_nodes, X=None, x_const=None):
    """layers: list of {"weights": [W0, W1(, W2)], "bias", "act"}; ``lap`` = (row_ptr, col, val) of L, or a list
    of such triples [T_1, T_2] for a model with three supports.  Returns scores[num_nodes, out]."""
    sups = [lap] if isinstance(lap, tuple) else list(lap)
    K = len(layers[0]["weights"])
    if len(sups) != K - 1:
        raise ValueError("model has %d supports, %d matrices given" % (K, len(sups)))
    L = len(layers)
    dims = np.array([layers[0]["weights"][0].shape[0]] + [l["weights"][0].shape[1] for l in layers], np.int32)
    cats = [np.ascontiguousarray(np.concatenate([np.asarray(w, np.float32) for w in l["weights"]], axis=1)) for l in layers]
    bs = [None if l.get("bias") is None else np.ascontiguousarray(l["bias"], np.float32).ravel() for l in layers]
    wptr = (C.c_void_p * L)(*[c.ctypes.data for c in cats])
    bptr = (C.c_void_p * L)(*[(b.ctypes.data if b is not None else None) for b in bs])
    acts = np.array([ACTS[l.get("act")] for l in layers], np.int32)
    if x_const is None:
        x_const = float(np.float32(1.0 / dims[0]))
    if X is not None:
        X = np.ascontiguousarray(X, np.float32)
    scores = np.empty((num_nodes, int(dims[-1])), np.float32)
    keep = [[np.ascontiguousarray(a) for a in s] for s in sups]
    rp = (C.c_void_p * (K - 1))(*[s[0].ctypes.data for s in keep])
    cl = (C.c_void_p * (K - 1))(*[s[1].ctypes.data for s in keep])
    vl = (C.c_void_p * (K - 1))(*[s[2].ctypes.data for s in keep])
    rc = lib().orc_forward_poly(num_nodes, K, rp, cl, vl, L, _p(dims), wptr, bptr, _p(acts), _p(X),
                                C.c_float(x_const), _p(scores))
    if rc:
        raise MemoryError("orc_forward")
    return scores


def lgs(graph_ptr, row_ptr, col_idx, prio, max_rounds=0, sum_weights=None, want_stats=True):
    graph_ptr = np.ascontiguousarray(graph_ptr, np.int32)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    prio = np.ascontiguousarray(prio, np.float64)
    B = graph_ptr.size - 1
    n = int(graph_ptr[-1])
    state = np.zeros(max(n, 1), np.uint8)
    rounds = np.zeros(max(B, 1), np.int32)
    stats = np.zeros((max(B, 1), 2), np.int64) if want_stats else None
    overhead = np.zeros(max(n, 1), np.int32) if want_stats else None
    totals = np.zeros(max(B, 1), np.float64)
    if sum_weights is not None:
        sum_weights = np.ascontiguousarray(sum_weights, np.float64)
    fault = lib().orc_lgs(B, _p(graph_ptr), _p(row_ptr), _p(col_idx), _p(prio), int(max_rounds), _p(state),
                          _p(rounds), _p(stats), _p(overhead), _p(sum_weights), _p(totals))
    return {"state": state[:n], "rounds": rounds[:B], "stats": None if stats is None else stats[:B],
            "overhead": None if overhead is None else overhead[:n], "totals": totals[:B], "fault": fault}


def solve(host_batch, layers, predict="mwis"):
    """Whole path on the CPU twin: supports -> forward -> priority -> lgs."""
    hb = host_batch
    lrp, lc, lv, fault = supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    sups = [(lrp, lc, lv)]
    if len(layers[0]["weights"]) == 3:
        r2, c2, v2, f2 = supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)
        sups.append((r2, c2, v2))
        fault |= f2
    scores = forward(sups, layers, hb.num_nodes)
    s = scores[:, 0]
    prio = s.astype(np.float64) * hb.weights if predict == "mwis" else s.astype(np.float64)
    out = lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, prio, sum_weights=hb.weights, want_stats=False)
    out["scores"] = scores
    out["fault"] |= fault
    return out
