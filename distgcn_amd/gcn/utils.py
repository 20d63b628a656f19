"""Host-side mirror of the reference's ``gcn/utils.py`` (the functions on the inference path).

These return the same host objects as the reference (SciPy matrices, ``(coords, values, shape)``
tuples) so callers that build a ``state`` by hand keep working.  The hot path does NOT go through
them: ``DQNAgent.solve_mwis`` / ``solve_mwis_batch`` upload the adjacency and build the support on
the device (``dgcn_supports_batch`` / ``dgcn_solve_batch``).  They are NumPy/SciPy because their
outputs ARE host objects.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def sparse_to_tuple(sparse_mx):
    """``(coords, values, shape)`` of a sparse matrix, or of each matrix of a list (``utils.py:79-95``)."""
    def one(mx):
        mx = mx if sp.isspmatrix_coo(mx) else mx.tocoo()
        return np.vstack((mx.row, mx.col)).transpose(), mx.data, mx.shape
    if isinstance(sparse_mx, list):
        return [one(m) for m in sparse_mx]
    return one(sparse_mx)


def preprocess_features(features):
    """Row-normalise, ``inf -> 0`` (``utils.py:98-106``)."""
    rowsum = np.array(features.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    return sparse_to_tuple(sp.diags(r_inv).dot(features))


def normalize_adj(adj):
    """``D^-1/2 A D^-1/2`` with 0 for isolated vertices (``utils.py:120-127``)."""
    adj = sp.coo_matrix(adj)
    rowsum = np.array(adj.sum(1))
    with np.errstate(divide="ignore"):
        d_inv_sqrt = np.power(rowsum, -0.5).flatten()
    d_inv_sqrt[np.isinf(d_inv_sqrt)] = 0.
    d = sp.diags(d_inv_sqrt)
    return adj.dot(d).transpose().dot(d).tocoo()


def simple_polynomials(adj, k):
    """``[I, L, L^2, ...]`` as tuples, ``L = I - normalize_adj(adj)`` (``utils.py:258-274``)."""
    n = adj.shape[0]
    lap = sp.eye(n) - normalize_adj(adj)
    t_k = [sp.eye(n), lap]
    for _ in range(2, k + 1):
        t_k.append(t_k[-1] * lap)
    return sparse_to_tuple(t_k)


def construct_feed_dict4pred(features, support, placeholders=None, adj_coo=(), hidden=None):
    """The reference maps TF placeholders to values (``utils.py:157-168``); without TF the "feed" is
    simply the state the models' ``predict`` takes."""
    return {"features": features, "support": support}
