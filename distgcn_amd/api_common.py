"""Glue shared by the drop-in modules (heuristics, gcn.*, mwis_*_call): the process-wide engine and
conversions between the reference's host objects (SciPy matrices, COO tuples) and device batches."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import _lib
from .batch import HostBatch

_engine = None


def get_engine(device=None):
    """The process-wide Engine (created on first use; raises DgcnError without a GPU - the product
    has no CPU fallback)."""
    global _engine
    if _engine is None or (device is not None and str(_engine.device) != str(device)):
        from .engine import Engine
        _engine = Engine(device or "cuda")
    return _engine


def as_csr(adj):
    """Canonical CSR (sorted rows, no duplicates) of any SciPy matrix / array.  A matrix that already is
    canonical CSR is returned AS IS (callers never modify the result in place)."""
    if sp.isspmatrix_csr(adj) and adj.has_canonical_format:
        a = adj
    else:
        a = sp.csr_matrix(adj)
        a.sum_duplicates()
        a.sort_indices()
    if a.shape[0] != a.shape[1]:
        raise ValueError("adjacency must be square, got %s" % (a.shape,))
    return a


def single_batch(adj, weights=None) -> HostBatch:
    a = as_csr(adj)
    w = None if weights is None else [np.asarray(weights, dtype=np.float64).ravel()]
    return HostBatch.from_csr_lists([a.indptr], [a.indices], w)


def tuple_to_dense(tup, dtype=np.float32):
    coords, values, shape = tup
    out = np.zeros(shape, dtype=dtype)
    coords = np.asarray(coords)
    if coords.size:
        out[coords[:, 0], coords[:, 1]] = np.asarray(values).astype(dtype)
    return out


def state_to_device(engine, state, input_dim):
    """Turn the reference's ``state`` dict into (DeviceBatch with its support attached, X or None).

    A state made by this package's ``makestate`` carries the adjacency (key ``"adj"``): the support
    is then built on the device.  A foreign state (supports computed elsewhere as COO tuples,
    ``gcn/utils.py:258-274``) is honoured as given: ``support[1]`` is uploaded as the CSR of L.
    """
    import torch
    feats = state["features"]
    n = int(feats[2][0])
    X = tuple_to_dense(feats, np.float32)
    if X.shape[1] != input_dim:
        raise ValueError("state features have %d columns, model expects %d" % (X.shape[1], input_dim))
    const = X[0, 0] if X.size else 0.0
    Xd = None if (X.size and np.all(X == const)) else torch.from_numpy(X).to(engine.device)
    x_const = float(const)
    adj = state.get("adj") if isinstance(state, dict) else None
    if adj is not None:
        db = engine.upload(single_batch(adj))
        return db, Xd, x_const
    sup = dict.__getitem__(state, "support") if isinstance(state, dict) else state["support"]
    if len(sup) not in (2, 3):
        raise _lib.DgcnError("[I, L] (max_degree=1) and [I, L, L.L] (max_degree=2) supports are implemented; state has %d"
                             % len(sup))

    def csr_of(tup):
        coords, values, shape = tup
        m = sp.csr_matrix((np.asarray(values, dtype=np.float64), (coords[:, 0], coords[:, 1])), shape=shape)
        m.sort_indices()
        return m

    def upload_csr(m):
        row_ptr = torch.from_numpy(m.indptr.astype(np.int32)).to(engine.device)
        col = torch.from_numpy(m.indices.astype(np.int32)).to(engine.device)
        val = torch.from_numpy(m.data.astype(np.float32)).to(engine.device)  # TF's float64 -> float32 feed cast
        csr = _lib.DgcnCsr(n, int(m.nnz), int(m.nnz), row_ptr.data_ptr(), col.data_ptr(), val.data_ptr())
        return {"row_ptr": row_ptr, "col_idx": col, "values": val, "c": csr}

    lap = csr_of(sup[1])
    off = lap.copy().tolil()
    off.setdiag(0)
    off = sp.csr_matrix(off)
    off.eliminate_zeros()
    db = engine.upload(HostBatch.from_csr_lists([off.indptr], [off.indices]))
    db.lap = upload_csr(lap)
    if len(sup) == 3:
        db.lap2 = upload_csr(csr_of(sup[2]))
    return db, Xd, x_const
