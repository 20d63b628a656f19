"""Device engine: thin Python over the C ABI (include/dgcn.h).

PyTorch is used for device memory and streams only; every computation on the path runs in
libdgcn.so.  All methods enqueue on the current torch stream and do not synchronise unless they
return host data.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

from . import _lib
from .batch import DeviceBatch, HostBatch

ACT_CODES = {"linear": _lib.ACT_LINEAR, "identity": _lib.ACT_LINEAR, None: _lib.ACT_LINEAR,
             "leaky_relu": _lib.ACT_LEAKY_RELU, "relu": _lib.ACT_RELU}

MODE_LAYERED, MODE_FUSED = 0, 1


def dinv_table(max_degree: int) -> np.ndarray:
    """float64 d^-1/2 for d = 0..max_degree with inf -> 0, computed with the very NumPy call the
    reference uses (gcn/utils.py:124-125) so the bits agree."""
    d = np.arange(max_degree + 1, dtype=np.float64)
    with np.errstate(divide="ignore"):
        t = np.power(d, -0.5)
    t[np.isinf(t)] = 0.0
    return t


_NP = {"uint8": np.uint8, "int32": np.int32, "int64": np.int64, "float32": np.float32, "float64": np.float64}


def packed_layout(specs):
    """(total bytes, {name: (byte offset, bytes, dtype name)}) of arrays packed into one byte buffer, widest
    element type first, every array 16-byte aligned.  specs: (name, element count, dtype name); count 0 = absent."""
    off, lay = 0, {}
    for name, count, dt in sorted(specs, key=lambda x: -np.dtype(_NP[x[2]]).itemsize):
        if count <= 0:
            continue
        nb = count * np.dtype(_NP[dt]).itemsize
        lay[name] = (off, nb, dt)
        off = (off + nb + 15) & ~15
    return max(off, 16), lay


def solve_buffer_specs(num_nodes: int, num_graphs: int, want_scores: bool):
    """The arrays of one rank's packed solve result (``Engine.solve_buffers``)."""
    return [("totals", num_graphs, "float64"), ("scores", num_nodes if want_scores else 0, "float32"),
            ("rounds", num_graphs, "int32"), ("status", 1, "int32"), ("state", num_nodes, "uint8")]


class DeviceModel:
    """Layer stack on the device.  ``layers`` is a list of dicts
    {"weights": [W_0, W_1] (each [in, out] float32), "bias": None | [out], "act": str}."""

    def __init__(self, layers: List[dict], device="cuda"):
        import torch
        if not layers:
            raise ValueError("model has no layers")
        self.device = torch.device(device)
        self.layers = layers
        self.num_supports = len(layers[0]["weights"])
        self._keep = []
        arr = (_lib.DgcnLayer * len(layers))()
        for i, lyr in enumerate(layers):
            ws = [np.asarray(w, dtype=np.float32) for w in lyr["weights"]]
            if len(ws) != self.num_supports:
                raise ValueError("layer %d has %d weight matrices, expected %d" % (i, len(ws), self.num_supports))
            cat = np.ascontiguousarray(np.concatenate(ws, axis=1))  # [in, K*out]
            wt = torch.from_numpy(cat).to(self.device)
            bt = None
            if lyr.get("bias") is not None:
                bt = torch.from_numpy(np.ascontiguousarray(lyr["bias"], dtype=np.float32).ravel()).to(self.device)
            self._keep += [wt, bt]
            arr[i] = _lib.DgcnLayer(ws[0].shape[0], ws[0].shape[1], wt.data_ptr(),
                                    bt.data_ptr() if bt is not None else None, ACT_CODES[lyr.get("act")])
        self._layer_arr = arr
        self.c = _lib.DgcnModel(len(layers), self.num_supports, arr)
        self.in_dim = int(arr[0].in_dim)
        self.out_dim = int(arr[len(layers) - 1].out_dim)


class Engine:
    def __init__(self, device="cuda"):
        import torch
        if not torch.cuda.is_available():
            raise _lib.DgcnError("no GPU visible: distgcn_amd runs on MI355X only (there is no CPU fallback)")
        self.torch = torch
        self.device = torch.device(device)
        self.lib = _lib.load()
        self._table = None
        self._ws = None

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dinv(self, max_degree: int):
        if self._table is None or self._table.numel() <= max_degree:
            n = max(1024, 2 * (max_degree + 1))
            self._table = self.torch.from_numpy(dinv_table(n - 1)).to(self.device)
        return self._table

    def _workspace(self, nbytes: int):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = self.torch.empty(max(nbytes, 256), dtype=self.torch.uint8, device=self.device)
        return self._ws

    _NP = _NP

    def _packed(self, specs):
        """Device arrays as views into ONE byte tensor, so a caller can fetch them all with a single
        device-to-host copy (``fetch_packed``).  specs: (name, element count, dtype name); count 0 = absent."""
        t = self.torch
        off, lay = packed_layout(specs)
        flat = t.zeros(off, dtype=t.uint8, device=self.device)
        out = {"flat": flat, "layout": lay}
        for name, _, _ in specs:
            if name in lay:
                o, nb, dt = lay[name]
                out[name] = flat[o:o + nb].view(getattr(t, dt))
            else:
                out[name] = None
        return out

    @classmethod
    def fetch_packed(cls, out):
        """One device-to-host copy -> {name: NumPy array} for everything ``_packed`` laid out."""
        host = out["flat"].cpu().numpy()
        return {name: host[o:o + nb].view(cls._NP[dt]) for name, (o, nb, dt) in out["layout"].items()}

    def upload(self, host: HostBatch) -> DeviceBatch:
        with self.torch.cuda.device(self.device):
            return DeviceBatch(host, self.device)

    def check_status(self, status) -> None:
        self.check_status_bits(int(status.item()))

    @staticmethod
    def check_status_bits(bits: int) -> None:
        if bits & 16:
            # the cluster variant of the fused kernel relies on a placement that this device / partition mode does not give:
            # switch it off for the rest of the process, so that a retry (and every later call) takes the ordinary launch
            _lib.load().dgcn_set_cluster(0)
        if bits:
            raise _lib.DgcnError("device-side validation failed: " + _lib.fault_text(bits))

    # ------------------------------------------------------------------ A1/A2
    def supports(self, b: DeviceBatch, status=None):
        """L = I - D^-1/2 A D^-1/2 for the batch (gcn/utils.py:258-274, k=1).  Cached on the batch."""
        t = self.torch
        if b.lap is not None:
            return b.lap
        n, e = b.host.num_nodes, b.host.num_edges
        row_ptr = t.empty(n + 1, dtype=t.int32, device=self.device)
        col = t.empty(max(n + e, 1), dtype=t.int32, device=self.device)
        val = t.empty(max(n + e, 1), dtype=t.float32, device=self.device)
        own = status is None
        if own:
            status = t.zeros(1, dtype=t.int32, device=self.device)
        tab = self._dinv(b.host.max_degree)
        if n == 0:
            row_ptr.zero_()
        _lib.check(self.lib.dgcn_supports_batch(C.byref(b.c), tab.data_ptr(), int(tab.numel()), row_ptr.data_ptr(),
                                                col.data_ptr(), val.data_ptr(), status.data_ptr(), self._stream()),
                   "dgcn_supports_batch")
        if own:
            self.check_status(status)
        csr = _lib.DgcnCsr(n, n + e, b.host.max_graph_edges + b.host.max_nodes, row_ptr.data_ptr(), col.data_ptr(), val.data_ptr())
        b.lap = {"row_ptr": row_ptr, "col_idx": col, "values": val, "c": csr}
        return b.lap

    def supports2(self, b: DeviceBatch, status=None):
        """T_2 = L.L for the batch, formed explicitly like the reference's ``t_k[-1]*laplacian``
        (gcn/utils.py:268-271); float32 values bit-identical to the reference's.  Cached on the batch.
        One device-to-host read of the entry count between the two passes (the caller allocates)."""
        t = self.torch
        if getattr(b, "lap2", None) is not None:
            return b.lap2
        n = b.host.num_nodes
        own = status is None
        if own:
            status = t.zeros(1, dtype=t.int32, device=self.device)
        tab = self._dinv(b.host.max_degree)
        row_ptr = t.zeros(n + 1, dtype=t.int32, device=self.device)
        _lib.check(self.lib.dgcn_supports2_count_batch(C.byref(b.c), tab.data_ptr(), int(tab.numel()), row_ptr.data_ptr(),
                                                       status.data_ptr(), self._stream()), "dgcn_supports2_count_batch")
        nnz = int(row_ptr[n].item())
        col = t.empty(max(nnz, 1), dtype=t.int32, device=self.device)
        val = t.empty(max(nnz, 1), dtype=t.float32, device=self.device)
        _lib.check(self.lib.dgcn_supports2_fill_batch(C.byref(b.c), tab.data_ptr(), int(tab.numel()), row_ptr.data_ptr(),
                                                      col.data_ptr(), val.data_ptr(), status.data_ptr(), self._stream()),
                   "dgcn_supports2_fill_batch")
        if own:
            self.check_status(status)
        csr = _lib.DgcnCsr(n, nnz, 0, row_ptr.data_ptr(), col.data_ptr(), val.data_ptr())
        b.lap2 = {"row_ptr": row_ptr, "col_idx": col, "values": val, "c": csr}
        return b.lap2

    # ------------------------------------------------------------------ K4
    def spmm(self, csr: dict, Z, C_feat: int, ldz: Optional[int] = None, graph_ptr=None, num_graphs=0, max_nodes=0,
             Y0=None, ldy0=0, bias=None, act="linear", out=None, precise=False):
        """``precise``: the row sums carried in double (``dgcn_spmm_f64acc_batch``, the contract of layer index 0)."""
        t = self.torch
        n = int(csr["c"].num_rows)
        ldz = ldz or int(Z.shape[-1])
        if out is None:
            out = t.empty((n, C_feat), dtype=t.float32, device=self.device)
        if precise:
            _lib.check(self.lib.dgcn_spmm_f64acc_batch(
                C.byref(csr["c"]), graph_ptr.data_ptr() if graph_ptr is not None else None, num_graphs, max_nodes,
                Z.data_ptr(), ldz, C_feat, Y0.data_ptr() if Y0 is not None else None, ldy0,
                bias.data_ptr() if bias is not None else None, ACT_CODES[act], out.data_ptr(), int(out.shape[-1]),
                self._stream()), "dgcn_spmm_f64acc_batch")
            return out
        _lib.check(self.lib.dgcn_spmm_batch(
            C.byref(csr["c"]), graph_ptr.data_ptr() if graph_ptr is not None else None, num_graphs, max_nodes,
            Z.data_ptr(), ldz, C_feat, Y0.data_ptr() if Y0 is not None else None, ldy0,
            bias.data_ptr() if bias is not None else None, ACT_CODES[act], out.data_ptr(), int(out.shape[-1]),
            self._stream()), "dgcn_spmm_batch")
        return out

    # ------------------------------------------------------------------ K2/K3
    def transform(self, H, W, rows: Optional[int] = None, h_const: float = 1.0, precise=False):
        """``precise``: the k chains carried in double (``dgcn_transform_f64acc_batch``, the contract of layer index 1)."""
        t = self.torch
        cin, ctot = int(W.shape[0]), int(W.shape[1])
        if H is not None:
            rows = int(H.shape[0])
        out = t.empty((rows, ctot), dtype=t.float32, device=self.device)
        fn = self.lib.dgcn_transform_f64acc_batch if precise else self.lib.dgcn_transform_batch
        _lib.check(fn(H.data_ptr() if H is not None else None,
                                                 int(H.shape[1]) if H is not None else cin, h_const, rows, cin,
                                                 W.data_ptr(), ctot, out.data_ptr(), ctot, self._stream()),
                   "dgcn_transform_batch")
        return out

    # ------------------------------------------------------------------ A4-A6
    def forward(self, b: DeviceBatch, model: DeviceModel, X=None, x_const: Optional[float] = None,
                mode: int = MODE_LAYERED, out=None):
        """scores[num_nodes, out_dim] (float32) = model.outputs for every node of the batch."""
        t = self.torch
        lap = self.supports(b)
        if x_const is None:
            x_const = float(np.float32(1.0 / model.in_dim))  # row-normalised all-ones features
        if out is None:
            out = t.empty((b.host.num_nodes, model.out_dim), dtype=t.float32, device=self.device)
        if model.num_supports == 3:  # [I, L, L.L]: layer by layer through the poly entry
            if mode != MODE_LAYERED:
                raise _lib.DgcnError("max_degree=2 models run layer by layer (mode=MODE_LAYERED)")
            lap2 = self.supports2(b)
            need = int(self.lib.dgcn_gcn_forward_workspace(C.byref(b.c), C.byref(model.c), 0))
            ws = self._workspace(need)
            sups = (C.POINTER(_lib.DgcnCsr) * 2)(C.pointer(lap["c"]), C.pointer(lap2["c"]))
            _lib.check(self.lib.dgcn_gcn_forward_poly_batch(C.byref(b.c), sups, C.byref(model.c),
                                                            X.data_ptr() if X is not None else None, x_const,
                                                            out.data_ptr(), ws.data_ptr(), int(ws.numel()), self._stream()),
                       "dgcn_gcn_forward_poly_batch")
            return out
        need = int(self.lib.dgcn_gcn_forward_workspace(C.byref(b.c), C.byref(model.c), mode))
        if need == 0:
            _lib.check(-1, "dgcn_gcn_forward_workspace")
        ws = self._workspace(need)
        _lib.check(self.lib.dgcn_gcn_forward_batch(C.byref(b.c), C.byref(lap["c"]), C.byref(model.c),
                                                   X.data_ptr() if X is not None else None, x_const, out.data_ptr(),
                                                   ws.data_ptr(), int(ws.numel()), mode, self._stream()),
                   "dgcn_gcn_forward_batch")
        return out

    def head_dual(self, b: DeviceBatch, act):
        """GCN2_DQN(is_dual=True) output head (gcn/models.py:651-653): [num_nodes, D] -> [num_nodes, D - 1]."""
        t = self.torch
        D = int(act.shape[1])
        out = t.empty((b.host.num_nodes, D - 1), dtype=t.float32, device=self.device)
        _lib.check(self.lib.dgcn_head_dual_batch(act.data_ptr(), D, b.graph_ptr.data_ptr(), b.host.num_graphs, out.data_ptr(),
                                                 self._stream()), "dgcn_head_dual_batch")
        return out

    def head_skip(self, act, kernel, bias, X=None, x_const: float = 1.0, in_dim: int = 1):
        """GCN_DQN(skip=True) output head (gcn/models.py:505-521): dense(concat([X, act])) -> [num_nodes, D]."""
        t = self.torch
        out = t.empty_like(act)
        _lib.check(self.lib.dgcn_head_skip_batch(X.data_ptr() if X is not None else None, x_const, in_dim, act.data_ptr(),
                                                 int(act.shape[1]), kernel.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 int(act.shape[0]), out.data_ptr(), self._stream()), "dgcn_head_skip_batch")
        return out

    def argmax(self, b: DeviceBatch, scores):
        t = self.torch
        out = t.empty(b.host.num_graphs, dtype=t.int32, device=self.device)
        _lib.check(self.lib.dgcn_argmax_batch(scores.data_ptr(), int(scores.shape[-1]) if scores.dim() > 1 else 1,
                                              b.graph_ptr.data_ptr(), b.host.num_graphs, out.data_ptr(),
                                              self._stream()), "dgcn_argmax_batch")
        return out

    # ------------------------------------------------------------------ A7-A9
    def lgs(self, b: DeviceBatch, prio=None, scores=None, weights=None, max_rounds: int = 0, want_stats=False,
            want_overhead=False, sum_weights=None, want_totals=True, status=None):
        """Local greedy search over the batch.  Returns a dict of device tensors."""
        n, B = b.host.num_nodes, b.host.num_graphs
        own = status is None
        pk = self._packed([("totals", max(B, 1) if want_totals else 0, "float64"),
                           ("stats", 2 * max(B, 1) if want_stats or want_overhead else 0, "int64"),
                           ("rounds", max(B, 1), "int32"), ("overhead", max(n, 1) if want_overhead else 0, "int32"),
                           ("status", 1 if own else 0, "int32"), ("state", max(n, 1), "uint8")])
        state, rounds, totals, overhead = pk["state"], pk["rounds"], pk["totals"], pk["overhead"]
        stats = pk["stats"].reshape(-1, 2) if pk["stats"] is not None else None
        if own:
            status = pk["status"]
        p = lambda x: x.data_ptr() if x is not None else None
        _lib.check(self.lib.dgcn_lgs_batch(C.byref(b.c), p(prio), p(scores), p(weights), int(max_rounds), p(state),
                                           p(rounds), p(stats), p(overhead), p(sum_weights), p(totals), p(status),
                                           self._stream()), "dgcn_lgs_batch")
        return {"state": state[:n], "rounds": rounds[:B], "stats": stats, "overhead": overhead,
                "totals": totals, "status": status, "flat": pk["flat"], "layout": pk["layout"]}

    def margin_risk(self, b: DeviceBatch, state, delta: float, prio=None, scores=None, weights=None):
        """per graph, the number of excluded vertices whose exclusion a score error of ``delta`` could
        overturn (``dgcn_margin_risk_batch``); 0 proves the graph's set is the same for every score vector within
        ``delta`` of this one.  -> int32 device tensor [num_graphs]."""
        t = self.torch
        out = t.zeros(max(b.host.num_graphs, 1), dtype=t.int32, device=self.device)
        p = lambda x: x.data_ptr() if x is not None else None
        _lib.check(self.lib.dgcn_margin_risk_batch(C.byref(b.c), p(prio), p(scores), p(weights), state.data_ptr(),
                                                   float(delta), out.data_ptr(), self._stream()), "dgcn_margin_risk_batch")
        return out[:b.host.num_graphs]

    def lgs_masked(self, b: DeviceBatch, prio, init_state, num_instances: int, sum_weights=None, max_rounds: int = 0,
                   prio_stride: int = 0):
        """Greedy search on ``num_instances`` residuals of the same batch (``dgcn_lgs_masked_batch``).
        ``init_state`` is uint8 ``[num_instances, num_nodes]`` (non-zero = vertex taken out)."""
        t = self.torch
        n, B = b.host.num_nodes, b.host.num_graphs
        state = t.empty((num_instances, max(n, 1)), dtype=t.uint8, device=self.device)
        rounds = t.empty((num_instances, max(B, 1)), dtype=t.int32, device=self.device)
        totals = t.empty((num_instances, max(B, 1)), dtype=t.float64, device=self.device)
        status = t.zeros(1, dtype=t.int32, device=self.device)
        _lib.check(self.lib.dgcn_lgs_masked_batch(C.byref(b.c), prio.data_ptr(), int(prio_stride), init_state.data_ptr(),
                                                  int(num_instances), int(max_rounds), state.data_ptr(), rounds.data_ptr(),
                                                  sum_weights.data_ptr() if sum_weights is not None else None,
                                                  totals.data_ptr(), status.data_ptr(), self._stream()),
                   "dgcn_lgs_masked_batch")
        return {"state": state[:, :n], "rounds": rounds[:, :B], "totals": totals[:, :B], "status": status}

    # ------------------------------------------------------------------ A10 (batched)
    def solve(self, b: DeviceBatch, model: DeviceModel, predict: str = "mwis", mode: int = MODE_LAYERED, X=None,
              x_const=None):
        """GCN forward -> priority -> local greedy for every graph of the batch
        (mwis_gdpg_call.py:200-235, batched).  ``b.weights`` must be set."""
        if b.weights is None:
            raise ValueError("batch has no vertex weights")
        if model.out_dim != 1:
            raise _lib.DgcnError("solve() needs a model with one output per node (diver_num=1)")
        status = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        if b.host.num_nodes == 0:  # nothing but empty graphs: rounds 0, totals 0 (heuristics.py loops do not run)
            t = self.torch
            B = b.host.num_graphs
            return {"state": t.empty(0, dtype=t.uint8, device=self.device),
                    "rounds": t.zeros(B, dtype=t.int32, device=self.device),
                    "totals": t.zeros(B, dtype=t.float64, device=self.device), "status": status,
                    "scores": t.empty((0, 1), dtype=t.float32, device=self.device), "stats": None, "overhead": None}
        if mode == MODE_FUSED:
            return self.solve_fused(b, model, predict=predict, X=X, x_const=x_const)
        self.supports(b, status=status)  # no host sync on the path: faults surface through res["status"]
        scores = self.forward(b, model, X=X, x_const=x_const, mode=mode)
        res = self.lgs(b, scores=scores, weights=b.weights if predict == "mwis" else None, sum_weights=b.weights,
                       status=status)
        res["scores"] = scores
        return res

    def solve_supported(self, b: DeviceBatch, model: DeviceModel) -> bool:
        """Does the per-graph fused kernel (one workgroup, one graph, everything in LDS) take this batch?"""
        return bool(self.lib.dgcn_solve_supported(C.byref(b.c), C.byref(model.c)))

    def solve_path(self, b: DeviceBatch, model: DeviceModel) -> int:
        """What ``solve_fused`` / ``solve_residual`` (``dgcn_solve_batch`` / ``dgcn_solve_residual_batch``) will run:
        1 = the fused kernels, 2 = the any-size device path (``csrc/general.hip``: graphs beyond 512 vertices or one
        CU's LDS, layer stacks wider than 32), 0 = neither."""
        return int(self.lib.dgcn_solve_path(C.byref(b.c), C.byref(model.c)))

    def solve_buffers(self, b: DeviceBatch, want_scores: bool = True, cap_nodes: int = 0, cap_graphs: int = 0):
        """Output buffers of ``solve_fused`` for a batch, for callers that re-use them across calls
        (a steady-state serving loop should not pay five allocations per batch).  All of them are views
        into ONE byte tensor, so everything comes back with a single device-to-host copy
        (``fetch_solve_buffers``) - or goes out in a single collective (``out["flat"]`` is what bench.py
        all-gathers: membership + totals + rounds + status of a rank in one buffer).  ``cap_nodes`` /
        ``cap_graphs``: lay the buffer out for that many vertices / graphs (every rank of a sharded batch
        uses the largest shard's sizes, so all ranks' buffers have one layout)."""
        n, B = max(b.host.num_nodes, cap_nodes, 1), max(b.host.num_graphs, cap_graphs, 1)
        out = self._packed(solve_buffer_specs(n, B, want_scores))
        if want_scores:
            out["scores"] = out["scores"].reshape(n, 1)
        return out

    @classmethod
    def fetch_solve_buffers(cls, out, num_nodes: int, num_graphs: int):
        """-> dict of NumPy arrays (state, totals, rounds, scores or None, status as int)."""
        h = cls.fetch_packed(out)
        return {"state": h["state"][:num_nodes], "totals": h["totals"][:num_graphs], "rounds": h["rounds"][:num_graphs],
                "scores": h["scores"][:num_nodes].reshape(-1, 1) if "scores" in h else None,
                "status": int(h["status"][0])}

    def solve_fused(self, b: DeviceBatch, model: DeviceModel, predict: str = "mwis", X=None, x_const=None,
                    want_scores: bool = True, out=None):
        """The whole path in one launch (dgcn_solve_batch): adjacency + weights in, membership out.
        ``out`` = buffers from ``solve_buffers`` to re-use (its status word accumulates faults)."""
        n, B = b.host.num_nodes, b.host.num_graphs
        if x_const is None:
            x_const = float(np.float32(1.0 / model.in_dim))
        tab = self._dinv(b.host.max_degree)
        if out is None:
            out = self.solve_buffers(b, want_scores)
        scores, state, rounds, totals, status = out["scores"], out["state"], out["rounds"], out["totals"], out["status"]
        p = lambda x: x.data_ptr() if x is not None else None
        need = int(self.lib.dgcn_solve_workspace(C.byref(b.c), C.byref(model.c)))
        ws = self._workspace(need)
        _lib.check(self.lib.dgcn_solve_batch(C.byref(b.c), C.byref(model.c), tab.data_ptr(), int(tab.numel()), p(X),
                                             x_const, p(b.weights), 1 if predict == "mwis" else 0, p(scores),
                                             p(state), p(rounds), p(totals), p(status), ws.data_ptr(), need,
                                             self._stream()),
                   "dgcn_solve_batch")
        return {"state": state[:n], "rounds": rounds[:B], "totals": totals[:B], "status": status,
                "scores": None if scores is None else scores[:n], "stats": None, "overhead": None}

    GREEDY_ROUNDS, GREEDY_CENTRAL, GREEDY_ROLLOUT = 0, 1, 2
    SCORES_GIVEN, COMPLETE_BY_PRIORITY, FINISH_SMALL = 1, 2, 4  # DGCN_RESIDUAL_* option bits

    def solve_residual(self, b: DeviceBatch, model: DeviceModel, state, predict: str = "mwis", greedy: int = 0,
                       max_rounds: int = 0, beam: int = 16, X=None, x_const=None, weight_features: bool = False,
                       want_scores: bool = False, max_steps: Optional[int] = None, out=None, options: int = 0,
                       scores=None, finish_small: Optional[bool] = None):
        """Iterative solvers on the device (dgcn_solve_residual_batch): repeat one launch per step on the
        residual graphs until no graph makes progress.  ``state`` (uint8 [num_nodes], 0 = undecided) is
        updated in place.  greedy = GREEDY_ROUNDS with max_rounds=1 is solve_mwis_dit, GREEDY_CENTRAL is
        solve_mwis_cit, GREEDY_ROLLOUT is solve_mwis_rollout (mwis_gdpg_call.py:278-318, 343-384, 596-659)
        for every graph of the batch at once.  -> {"state", "steps", "status", "scores" (last step)}
        ``finish_small`` (default: on for a search run to its end, off when ``max_steps`` asks for single steps): a graph
        with at most 64 undecided vertices runs the rest of its search inside the call that finds it so
        (DGCN_RESIDUAL_FINISH_SMALL, csrc/tail.hip); same final states, fewer calls - ``steps`` counts calls."""
        t = self.torch
        n, B = b.host.num_nodes, b.host.num_graphs
        if x_const is None:
            x_const = float(np.float32(1.0 / model.in_dim))
        tab = self._dinv(b.host.max_degree)
        if out is None:
            out = self.solve_buffers(b, want_scores)
        if options & self.SCORES_GIVEN:
            if scores is None:
                raise ValueError("options & SCORES_GIVEN needs scores")
            out = dict(out, scores=scores)
        if finish_small is None:
            finish_small = max_steps is None
        p = lambda x: x.data_ptr() if x is not None else None
        need = int(self.lib.dgcn_solve_workspace(C.byref(b.c), C.byref(model.c)))
        ws = self._workspace(need)
        steps = 0
        limit = max_steps if max_steps is not None else max(b.host.max_nodes, 1) + 1
        # Steps are launched a few at a time, each with its own progress word, and the words are read back together: a
        # step that decides nothing leaves the state as it is, so every later step of the group decides nothing either
        # (and costs next to nothing: a graph with nothing left returns at once) - one host round trip per group, not per step.
        group = 4
        issued = 0
        done = n == 0
        # (the status word is sticky - kernels OR into it, nobody clears it: only bits set DURING this search end it, a fault
        # an earlier call left in a re-used `out` is the caller's to read, not a reason to cut this search short)
        status_at_entry = out["status"].reshape(-1)[:1].to(t.int32).clone()
        while not done and steps < limit:
            k = min(group, limit - steps)
            progress = t.zeros(k, dtype=t.int32, device=self.device)
            for i in range(k):
                # (the tail's launch leaves at once while a graph of the batch is still large: asked for with every second
                # call, it costs a search ~3 us per step and starts at most one step late)
                opts = options | (self.FINISH_SMALL if finish_small and ((issued + i) & 1) else 0)
                _lib.check(self.lib.dgcn_solve_residual_batch(
                    C.byref(b.c), C.byref(model.c), tab.data_ptr(), int(tab.numel()), p(X), x_const,
                    1 if weight_features else 0, p(b.weights), 1 if predict == "mwis" else 0, int(greedy),
                    int(max_rounds), int(beam), int(opts), p(out["scores"]), state.data_ptr(), p(out["rounds"]), p(out["totals"]),
                    progress.data_ptr() + 4 * i, out["status"].data_ptr(), ws.data_ptr(), need, self._stream()),
                    "dgcn_solve_residual_batch")
            issued += k
            if max_steps == 1:  # a caller that asks for exactly one step does not need to know whether it decided anything
                steps = 1
                break
            # (the status word rides along: a fault such as a NaN priority leaves its graph undecided - "active" - for ever,
            # and the any-size path counts active graphs as progress; the caller's check_status reports it)
            words = t.cat([progress, out["status"].reshape(-1)[:1].to(t.int32), status_at_entry]).cpu().tolist()
            for v in words[:k]:
                if v == 0:
                    done = True
                    break
                steps += 1
            fresh = words[k] & ~words[k + 1]
            if fresh == 16:
                # a placement fault of the several-workgroups-per-graph launch (automatic for searches on small batches): the
                # step left the searches it touched as they were (fused.hip) - switch the variant off, clear the bit, go on
                self.lib.dgcn_set_cluster(0)
                out["status"].bitwise_and_(~16)
                done = False
            elif fresh:
                done = True
            group = min(2 * group, 32)  # (a search of ~100 steps: 6 read-backs instead of 14; at most 31 empty launches at its end)
        return {"state": state, "steps": steps, "status": out["status"],
                "scores": None if out["scores"] is None else out["scores"][:n]}

    # ------------------------------------------------------------------ timing hooks (bench.py)
    def timing(self, on: bool):
        """on=True clears earlier records and starts recording; on=False stops (records stay readable)."""
        if on:
            self.lib.dgcn_timing_reset()
        # (DGCN_TIMING_EVERY=N: an event pair round every N-th launch only - an A/B switch for what the instrumentation costs;
        # anything below 1 means 1: "on" never silently turns the recording off)
        self.lib.dgcn_timing_enable(max(1, int(os.environ.get("DGCN_TIMING_EVERY", "1"))) if on else 0)

    def timing_read(self, family: str):
        """(milliseconds, launches) of a kernel family since timing(True).  With every N-th launch sampled both are scaled by N
        (estimates of the whole region: the average per launch is unchanged, per-step figures stay per step)."""
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        _lib.check(self.lib.dgcn_timing_read(family.encode(), C.byref(ms), C.byref(n)), "dgcn_timing_read")
        every = max(1, int(self.lib.dgcn_timing_sampling()))
        return ms.value * every, n.value * every
