"""Block-diagonal CSR batches of conflict graphs (host build + device residency).

The reference handles one SciPy matrix per call (``mwis_dqn_call.py:198-261``); here many graphs are
packed into one block-diagonal CSR so a single launch covers the whole batch.  Layout (see
``include/dgcn.h``): ``graph_ptr[B+1]`` node offsets, ``row_ptr[N+1]``, ``col_idx[nnz]`` with GLOBAL
node ids, int32 everywhere; adjacency values are implicit 1.0.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import _lib


def _pyptr():
    """The CPython pointer-table helper (csrc/pyptr.c), or None when it has not been built."""
    try:
        from . import _pyptr as m
        return m
    except ImportError:
        return None


def _addresses(arrays, kind):
    """(uint64 addresses, int64 element counts, itemsize) of a sequence of contiguous arrays."""
    n = len(arrays)
    addr = np.empty(max(n, 1), dtype=np.uint64)
    cnt = np.empty(max(n, 1), dtype=np.int64)
    m = _pyptr()
    if m is not None:
        isz = m.addresses(arrays, addr, cnt, kind)
        return addr, cnt, isz
    isz = 0  # interpreter fallback (~1 us per array); same checks
    for i, a in enumerate(arrays):
        if a is None:
            addr[i], cnt[i] = 0, 0
            continue
        if not isinstance(a, np.ndarray):  # (a converted temporary would be gone before its address is used)
            raise TypeError("item %d is a %s, expected a NumPy array" % (i, type(a).__name__))
        want_float = kind == 64
        ok = a.flags.c_contiguous and ((a.dtype == np.float64) if want_float else (a.dtype.kind == "i" and a.dtype.itemsize in (4, 8)))
        if ok and not want_float:
            isz = isz or (kind if kind in (4, 8) else a.dtype.itemsize)
            ok = a.dtype.itemsize == isz
        if not ok:
            raise TypeError("item %d has dtype %s, expected %s" % (i, a.dtype, "float64" if want_float else "int32 or int64 (one width)"))
        addr[i], cnt[i] = a.__array_interface__["data"][0], a.size
    return addr, cnt, (8 if kind == 64 else (isz or 4))


def pack_csr_lists(indptrs: Sequence[np.ndarray], indices: Sequence[np.ndarray], weights: Optional[Sequence[np.ndarray]] = None,
                   staging: Optional[np.ndarray] = None, alloc=None, threads: int = 0):
    """Native packing of per-graph CSR arrays into one block-diagonal batch (``dgcn_pack_batch``, include/dgcn.h).
    ``staging``: a uint8 array to write into (e.g. a view of pinned memory), or ``alloc(nbytes) -> uint8 array``,
    or neither (plain NumPy memory).  Arrays must be contiguous, indptr / indices all int32 or all int64, weights
    float64 (TypeError otherwise - the caller falls back to the NumPy path).  -> (staging, DgcnPackInfo)"""
    import ctypes as C
    lib = _lib.load()
    B = len(indptrs)
    if len(indices) != B or (weights is not None and len(weights) != B):
        raise ValueError("indptr / indices / weights lists differ in length")
    ap, cp, isz = _addresses(indptrs, 0)
    if B and int(cp[:B].min()) < 1:
        raise ValueError("an indptr array is empty")
    ai, ci, _ = _addresses(indices, isz)
    nn = (cp[:B] - 1).astype(np.int32)
    aw = None
    if weights is not None:
        aw, cw, _ = _addresses(weights, 64)
        if B and not np.array_equal(cw[:B], nn):
            raise ValueError("a weights array does not match its graph's vertex count")
    info = _lib.DgcnPackInfo()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    nnz = np.empty(max(B, 1), dtype=np.int64)
    _lib.check(lib.dgcn_pack_measure(vp(ap), vp(nn), B, isz, 1 if weights is not None else 0, C.byref(info), vp(nnz)),
               "dgcn_pack_measure")
    if B and not np.array_equal(ci[:B], nnz[:B]):
        raise ValueError("an indices array does not match its indptr")
    need = int(info.total_bytes)
    if staging is None:
        staging = alloc(need) if alloc is not None else np.empty(need, dtype=np.uint8)
    if staging.nbytes < need:
        raise ValueError("staging buffer of %d bytes is too small (%d needed)" % (staging.nbytes, need))
    _lib.check(lib.dgcn_pack_batch(vp(ap), vp(ai), vp(aw) if aw is not None else None, vp(nn), B, isz,
                                   staging.ctypes.data_as(C.c_void_p), staging.nbytes, C.byref(info), threads),
               "dgcn_pack_batch")
    return staging, info


class HostBatch:
    """Host-side block-diagonal CSR (NumPy only; no GPU needed)."""

    def __init__(self, graph_ptr, row_ptr, col_idx, weights=None):
        self.graph_ptr = np.ascontiguousarray(graph_ptr, dtype=np.int32)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.num_graphs = int(self.graph_ptr.size - 1)
        self.num_nodes = int(self.graph_ptr[-1]) if self.graph_ptr.size else 0
        self.num_edges = int(self.row_ptr[-1]) if self.row_ptr.size else 0
        if self.row_ptr.size != self.num_nodes + 1:
            raise ValueError("row_ptr has %d entries, expected %d" % (self.row_ptr.size, self.num_nodes + 1))
        if self.col_idx.size != self.num_edges:
            raise ValueError("col_idx has %d entries, row_ptr says %d" % (self.col_idx.size, self.num_edges))
        sizes = np.diff(self.graph_ptr)
        self.max_nodes = int(sizes.max()) if sizes.size else 0
        gedges = self.row_ptr[self.graph_ptr[1:]] - self.row_ptr[self.graph_ptr[:-1]] if self.num_graphs else np.zeros(0)
        self.max_graph_edges = int(gedges.max()) if self.num_graphs else 0
        deg = np.diff(self.row_ptr)
        self.max_degree = int(deg.max()) if deg.size else 0
        if self.weights is not None and self.weights.size != self.num_nodes:
            raise ValueError("weights has %d entries, expected %d" % (self.weights.size, self.num_nodes))

    @classmethod
    def from_packed(cls, staging: np.ndarray, info) -> "HostBatch":
        """Views into a buffer ``pack_csr_lists`` filled (no copy, no re-validation: the packer checked)."""
        self = cls.__new__(cls)
        B, n, e = int(info.num_graphs), int(info.num_nodes), int(info.num_edges)
        cut = lambda off, count, dt: staging[off:off + count * np.dtype(dt).itemsize].view(dt)
        self.graph_ptr = cut(int(info.off_graph_ptr), B + 1, np.int32)
        self.row_ptr = cut(int(info.off_row_ptr), n + 1, np.int32)
        self.col_idx = cut(int(info.off_col_idx), e, np.int32)
        self.weights = cut(int(info.off_weights), n, np.float64) if int(info.off_weights) >= 0 else None
        self.num_graphs, self.num_nodes, self.num_edges = B, n, e
        self.max_nodes, self.max_graph_edges, self.max_degree = int(info.max_nodes), int(info.max_graph_edges), int(info.max_degree)
        self._staging, self._info = staging, info
        return self

    @staticmethod
    def from_csr_lists(indptrs: Sequence[np.ndarray], indices: Sequence[np.ndarray],
                       weights: Optional[Sequence[np.ndarray]] = None, staging=None, alloc=None) -> "HostBatch":
        """One block-diagonal batch from per-graph CSR arrays.  Packed natively (``pack_csr_lists``) when the
        arrays qualify; otherwise (mixed index widths, non-contiguous views, float32 weights ...) through NumPy."""
        try:
            w = None if weights is None else [x if (isinstance(x, np.ndarray) and x.ndim == 1) else np.ascontiguousarray(x, dtype=np.float64).ravel()
                                              for x in weights]
            return HostBatch.from_packed(*pack_csr_lists(indptrs, indices, w, staging=staging, alloc=alloc))
        except (TypeError, BufferError):  # mixed index widths, non-contiguous or non-array inputs: the NumPy packer takes those
            pass
        B = len(indptrs)
        sizes = np.fromiter((p.size - 1 for p in indptrs), dtype=np.int64, count=B)
        nnzs = np.fromiter((p[-1] for p in indptrs), dtype=np.int64, count=B)
        graph_ptr = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(sizes, out=graph_ptr[1:])
        edge_ptr = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nnzs, out=edge_ptr[1:])
        if graph_ptr[-1] >= 2 ** 31 or edge_ptr[-1] >= 2 ** 31 - graph_ptr[-1]:
            raise ValueError("batch too large for int32 indices")
        if any(c.size != k for c, k in zip(indices, nnzs)):
            raise ValueError("an indices array does not match its indptr")
        # one concatenate + one offset add per array (no per-graph copies: packing 500 graphs is host time the
        # GPU path cannot hide)
        row_ptr = np.empty(graph_ptr[-1] + 1, dtype=np.int32)
        row_ptr[0] = 0
        if B:
            row_ptr[1:] = np.concatenate([p[1:] for p in indptrs]) + np.repeat(edge_ptr[:-1].astype(np.int32), sizes)
            col_idx = np.concatenate(indices).astype(np.int32, copy=False)
            col_idx += np.repeat(graph_ptr[:-1].astype(np.int32), nnzs)
        else:
            col_idx = np.empty(0, dtype=np.int32)
        w = None
        if weights is not None:
            w = np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in weights]) if len(weights) else np.zeros(0)
        return HostBatch(graph_ptr, row_ptr, col_idx, w)

    @staticmethod
    def from_scipy(adjs: Sequence, weights: Optional[Sequence] = None) -> "HostBatch":
        """Pack SciPy matrices (any format; ``loadmat`` gives COO/CSC) into one batch."""
        ps, cs = [], []
        for a in adjs:
            if not (sp.isspmatrix_csr(a) and a.has_canonical_format):
                a = sp.csr_matrix(a)
                a.sum_duplicates()
                a.sort_indices()
            if a.shape[0] != a.shape[1]:
                raise ValueError("adjacency must be square")
            ps.append(a.indptr)
            cs.append(a.indices)
        return HostBatch.from_csr_lists(ps, cs, weights)

    def graph_slices(self):
        return [(int(self.graph_ptr[g]), int(self.graph_ptr[g + 1])) for g in range(self.num_graphs)]

    def subset(self, lo: int, hi: int) -> "HostBatch":
        """Graphs [lo, hi) as their own batch (used to shard a batch over ranks)."""
        n0, n1 = int(self.graph_ptr[lo]), int(self.graph_ptr[hi])
        e0, e1 = int(self.row_ptr[n0]), int(self.row_ptr[n1])
        w = None if self.weights is None else self.weights[n0:n1]
        return HostBatch(self.graph_ptr[lo:hi + 1] - n0, self.row_ptr[n0:n1 + 1] - e0, self.col_idx[e0:e1] - n0, w)

    def select(self, ids) -> "HostBatch":
        """The graphs ``ids`` (any order) as a new batch."""
        ids = np.asarray(ids, dtype=np.int64)
        ps, cs, ws = [], [], []
        for g in ids:
            n0, n1 = int(self.graph_ptr[g]), int(self.graph_ptr[g + 1])
            e0, e1 = int(self.row_ptr[n0]), int(self.row_ptr[n1])
            ps.append(self.row_ptr[n0:n1 + 1].astype(np.int64) - e0)
            cs.append(self.col_idx[e0:e1].astype(np.int64) - n0)
            if self.weights is not None:
                ws.append(self.weights[n0:n1])
        return HostBatch.from_csr_lists(ps, cs, ws if self.weights is not None else None)

    def size_buckets(self, lds_budget: int = 80 * 1024, hidden: int = 32):
        """A launch of the fused kernel sizes LDS for its largest graph (``csrc/fused.hip``), so a FEW big
        graphs would halve (or worse) the residency of many small ones: then the two groups go in separate
        launches.  When the big graphs are a sizeable share of the batch the launch time is their latency
        anyway and one launch is faster (measured on the BA test2 mix: 477 us vs 497 us), so the batch is
        only split when they are under a quarter of it.  Returns a list of index arrays."""
        if self.num_graphs == 0:
            return [np.zeros(0, np.int64)]
        sizes = np.diff(self.graph_ptr).astype(np.int64)
        nnz = (self.row_ptr[self.graph_ptr[1:]] - self.row_ptr[self.graph_ptr[:-1]]).astype(np.int64)
        need = np.maximum(sizes, 64) * hidden * 4 * 2 + (nnz + 2 * sizes + 6) * 6 + sizes * 6 + 64
        small = np.flatnonzero(need <= lds_budget)
        big = np.flatnonzero(need > lds_budget)
        if min(small.size, big.size) < 32 or big.size * 4 > self.num_graphs:
            return [np.arange(self.num_graphs)]
        return [small, big]

    def scipy_graph(self, g: int):
        n0, n1 = int(self.graph_ptr[g]), int(self.graph_ptr[g + 1])
        e0, e1 = int(self.row_ptr[n0]), int(self.row_ptr[n1])
        n = n1 - n0
        return sp.csr_matrix((np.ones(e1 - e0), self.col_idx[e0:e1] - n0, self.row_ptr[n0:n1 + 1] - e0), shape=(n, n))


class DeviceBatch:
    """A HostBatch resident in HBM, plus the C struct the library takes."""

    def __init__(self, host: HostBatch, device="cuda", flat_src=None, dst=None):
        """``flat_src``: a (pinned) uint8 torch tensor that aliases ``host``'s staging buffer; ``dst``: a device uint8
        tensor to copy into instead of allocating (a serving loop re-uses both)."""
        import torch
        self.host = host
        self.device = torch.device(device)
        # one host-to-device copy for the whole batch: [graph_ptr | row_ptr | col_idx | weights], 16-byte aligned
        info = getattr(host, "_info", None)
        if info is not None and (host.weights is not None) == (int(info.off_weights) >= 0):
            # packed natively (dgcn_pack_batch): the staging buffer already IS that layout
            total = int(info.total_bytes)
            src = flat_src if flat_src is not None else torch.from_numpy(host._staging[:total])
            self._flat = torch.empty(total, dtype=torch.uint8, device=self.device) if dst is None else dst[:total]
            self._flat.copy_(src[:total], non_blocking=True)
            offs = {"graph_ptr": (int(info.off_graph_ptr), (host.num_graphs + 1) * 4),
                    "row_ptr": (int(info.off_row_ptr), (host.num_nodes + 1) * 4),
                    "col_idx": (int(info.off_col_idx), max(host.num_edges, 1) * 4)}
            if host.weights is not None:
                offs["weights"] = (int(info.off_weights), max(host.num_nodes, 1) * 8)
        else:
            parts = [("graph_ptr", host.graph_ptr), ("row_ptr", host.row_ptr),
                     ("col_idx", host.col_idx if host.col_idx.size else np.zeros(1, dtype=np.int32))]
            if host.weights is not None:
                parts.append(("weights", host.weights if host.weights.size else np.zeros(1, dtype=np.float64)))
            offs, off = {}, 0
            for name, arr in parts:
                offs[name] = (off, arr.nbytes)
                off = (off + arr.nbytes + 15) & ~15
            staging = np.empty(max(off, 16), dtype=np.uint8)
            for name, arr in parts:
                o, nb = offs[name]
                staging[o:o + nb] = arr.view(np.uint8).reshape(-1)
            self._flat = torch.from_numpy(staging).to(self.device, non_blocking=True)
        cut = lambda name, dt: self._flat[offs[name][0]:offs[name][0] + offs[name][1]].view(dt)
        self.graph_ptr = cut("graph_ptr", torch.int32)
        self.row_ptr = cut("row_ptr", torch.int32)
        self.col_idx = cut("col_idx", torch.int32)[:max(host.num_edges, 1)]
        self.weights = cut("weights", torch.float64)[:host.num_nodes] if host.weights is not None else None
        self.c = _lib.DgcnBatch(host.num_graphs, host.num_nodes, host.num_edges, host.max_nodes,
                                host.max_graph_edges, self.graph_ptr.data_ptr(), self.row_ptr.data_ptr(),
                                self.col_idx.data_ptr())
        self.lap = None  # filled by Engine.supports()
        self.lap2 = None  # filled by Engine.supports2() (max_degree = 2)

    @property
    def num_graphs(self):
        return self.host.num_graphs

    @property
    def num_nodes(self):
        return self.host.num_nodes
