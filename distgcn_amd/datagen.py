"""Seeded synthetic conflict graphs with the distributions of the reference's datasets.

The reference generates its data unseeded with NetworkX (``Data_Generation.py:46-58`` ER via
``fast_gnp_random_graph``; ``:83-95`` BA via ``barabasi_albert_graph(N, round(N*p))``; weights
``uniform(0,1)``, ``:48-50``).  The data folders do not travel to the GPU box, so benchmarks and
full-size tests use these generators: same distributions, ``numpy.random.default_rng(seed)`` with
seed = 20230600 + graph index.
"""
from __future__ import annotations

import numpy as np

from .batch import HostBatch

SEED0 = 20230600


def er_graph(n: int, p: float, rng) -> tuple:
    """G(n, p): undirected, no self-loops.  Returns CSR (indptr, indices) with sorted rows."""
    upper = np.triu(rng.random((n, n)) < p, k=1)
    a = upper | upper.T
    rows, cols = np.nonzero(a)  # row-major => sorted rows
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    return indptr, cols.astype(np.int64)


def ba_graph(n: int, m: int, rng) -> tuple:
    """Barabasi-Albert preferential attachment, m edges per new vertex (star seed on m+1 vertices,
    as NetworkX >= 2 does).  Returns CSR (indptr, indices) with sorted rows."""
    m = max(1, min(int(m), n - 1))
    src = list(range(1, m + 1))
    dst = [0] * m
    repeated = src + dst  # each vertex appears once per incident edge
    for v in range(m + 1, n):
        targets = set()
        while len(targets) < m:
            targets.add(repeated[int(rng.integers(len(repeated)))])
        for t in targets:
            src.append(v)
            dst.append(t)
        repeated.extend(targets)
        repeated.extend([v] * m)
    s = np.array(src + dst, dtype=np.int64)
    d = np.array(dst + src, dtype=np.int64)
    order = np.lexsort((d, s))
    s, d = s[order], d[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(s, minlength=n), out=indptr[1:])
    return indptr, d


def er_batch(num_graphs: int, n: int, p: float, seed0: int = SEED0, first_index: int = 0) -> HostBatch:
    ps, cs, ws = [], [], []
    for g in range(first_index, first_index + num_graphs):
        rng = np.random.default_rng(seed0 + g)
        ip, ix = er_graph(n, p, rng)
        ps.append(ip)
        cs.append(ix)
        ws.append(rng.random(n))
    return HostBatch.from_csr_lists(ps, cs, ws)


# the test2 recipe: N x average-degree grid (bash/run_data_generation.sh:22-26), 20 graphs per cell
TEST2_SIZES = (100, 150, 200, 250, 300)
TEST2_DEGREES = (2, 5, 10, 15, 20)


def ba_test2_batch(num_graphs: int, seed0: int = SEED0, first_index: int = 0) -> HostBatch:
    """``num_graphs`` BA graphs cycling through the test2 (N, m) grid; m = round(N*p) with
    p = avg_degree / N, i.e. m = avg_degree (Data_Generation.py:83-84)."""
    ps, cs, ws = [], [], []
    cells = [(n, d) for n in TEST2_SIZES for d in TEST2_DEGREES]
    for g in range(first_index, first_index + num_graphs):
        n, m = cells[g % len(cells)]
        rng = np.random.default_rng(seed0 + 1_000_000 + g)
        ip, ix = ba_graph(n, m, rng)
        ps.append(ip)
        cs.append(ix)
        ws.append(rng.random(n))
    return HostBatch.from_csr_lists(ps, cs, ws)


def random_model(num_layer: int, hidden: int, feature_size: int = 1, out_dim: int = 1, bias: bool = False,
                 last_act: str = "linear", seed: int = 7, num_supports: int = 2):
    """Glorot-uniform random layer stack of the GCN_DQN shape (gcn/inits.py glorot; models.py:536-573).
    Used when no checkpoint is available (benchmarks say so in their ``data`` field)."""
    rng = np.random.default_rng(seed)
    dims = [feature_size] + [hidden] * (num_layer - 1) + [out_dim]
    layers = []
    for i in range(num_layer):
        fan_in, fan_out = dims[i], dims[i + 1]
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        ws = [rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32) for _ in range(num_supports)]
        b = rng.uniform(-0.1, 0.1, size=fan_out).astype(np.float32) if bias else None
        act = last_act if i == num_layer - 1 else "leaky_relu"
        layers.append({"weights": ws, "bias": b, "act": act})
    return layers


def multichannel_batch(count, nflows, p, first_index=0, n_ch=3, keep=0.8):
    """Joint multi-channel conflict graphs as the reference's multi-channel scripts build them
    (wireless_dqn_test_mc.py:159-161): a single-channel conflict graph (stand-in: ER(nflows, p), the topology generator
    ``graph_util`` is absent from the reference), ``n_ch`` per-channel copies with every edge kept with probability ``keep``
    (``multichannel_conflict_simulate``, wireless_rollout_test_flood.py:83-95) and the joint graph on ``n_ch * nflows``
    vertices - the channels' graphs on the diagonal blocks plus a clique over every flow's copies (``:98-133``)."""
    import scipy.sparse as sp
    from . import wireless
    ps, cs, ws = [], [], []
    for g in range(first_index, first_index + count):
        rng = np.random.default_rng(SEED0 + 2_000_000 + g)
        ip, ix = er_graph(nflows, p, rng)
        base = sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(nflows, nflows))
        chans = wireless.multichannel_conflict_simulate(base, k=n_ch, p=keep, rng=np.random.RandomState(SEED0 % 100000 + g))
        _, joint = wireless.multichannel_conflict_graph(chans)
        ps.append(joint.indptr.astype(np.int64))
        cs.append(joint.indices.astype(np.int64))
        ws.append(rng.random(n_ch * nflows))
    return HostBatch.from_csr_lists(ps, cs, ws)
