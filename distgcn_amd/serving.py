"""Host-to-host serving loop: CSR lists in host memory -> membership / totals / rounds in host memory.

The reference's evaluation loop handles one graph at a time: load, ``makestate``, ``sess.run``, greedy, ratio
(``mwis_dqn_test.py:304-321``).  Here a batch goes through four stages that overlap across batches:

    pack (host threads, ``dgcn_pack_batch`` straight into pinned memory)
      -> one host-to-device copy -> ONE fused launch (``dgcn_solve_batch``) -> one device-to-host copy

``SolvePipeline`` keeps ``depth`` slots (pinned staging, device buffers, a stream and an event each); while the
GPU works on slot k the host packs slot k+1.  Only shapes the fused kernel takes are served here
(``Engine.solve_supported``); other shapes go through ``mwis_dqn_call.solve_host_batch``.

``HostSolver`` is the same loop behind two native calls (``dgcn_host_solver_submit`` / ``_result``, csrc/host_solver.hip):
what a per-graph caller - ``solve_mwis(adj, weights)``, one ``sess.run`` per graph in the reference - goes through, with no
framework call between the pack and the wait.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .batch import pack_csr_lists
from .engine import Engine, DeviceModel, packed_layout, solve_buffer_specs, _NP


class _Slot:
    def __init__(self, torch, device):
        self.torch, self.device = torch, device
        self.stream = torch.cuda.Stream(device=device)
        self.event = torch.cuda.Event()
        self.staging = None      # pinned uint8 tensor + its NumPy view
        self.staging_np = None
        self.dev = None          # device copy of the packed batch
        self.out = None          # device result buffer (packed: totals | rounds | status | state)
        self.res = None          # pinned host copy of `out`
        self.res_np = None
        self.ws = None
        self.cap = (0, 0)
        self.layout = None
        self.busy = False
        self.shape = None

    def ensure_staging(self, nbytes):
        if self.staging is None or self.staging.numel() < nbytes:
            t = self.torch
            nbytes = int(nbytes * 1.25) + 4096
            self.staging = t.empty(nbytes, dtype=t.uint8, pin_memory=True)
            self.staging_np = self.staging.numpy()
            self.dev = t.empty(nbytes, dtype=t.uint8, device=self.device)

    def ensure_out(self, n, B, want_scores=False):
        if n > self.cap[0] or B > self.cap[1] or self.out is None:
            t = self.torch
            cap = (max(int(n * 1.25), 64), max(int(B * 1.25), 8))
            size, self.layout = packed_layout(solve_buffer_specs(cap[0], cap[1], want_scores))
            # zero-filled ON THE SLOT'S STREAM: everything that touches the buffer afterwards is enqueued there, and the
            # stream is non-blocking - a fill on the current stream could land after (or in the middle of) the first solve
            with t.cuda.stream(self.stream):
                self.out = t.zeros(size, dtype=t.uint8, device=self.device)
            self.res = t.empty(size, dtype=t.uint8, pin_memory=True)
            self.res_np = self.res.numpy()
            self.cap = cap


class SolvePipeline:
    def __init__(self, engine: Engine, model: DeviceModel, depth: int = 2, predict: str = "mwis", pack_threads: int = 0,
                 want_scores: bool = False):
        self.eng, self.model, self.predict = engine, model, predict
        self.want_scores = want_scores
        self.torch = engine.torch
        self.lib = engine.lib
        self.pack_threads = pack_threads
        self.slots = [_Slot(self.torch, engine.device) for _ in range(max(1, depth))]
        self.k = 0
        self.x_const = float(np.float32(1.0 / model.in_dim))
        self.table = engine._dinv(4095)  # one d^-1/2 table for every slot (never re-allocated while streams run)

    def _pack(self, slot: _Slot, indptrs, indices, weights):
        """Host stage: the batch into the slot's pinned staging buffer (``dgcn_pack_batch``; releases the GIL)."""
        if slot.staging is None:
            guess = sum(int(p.size) for p in indptrs) * 12 + sum(int(c.size) for c in indices) * 4 + 4096
            slot.ensure_staging(guess)
        while True:
            try:
                _, info = pack_csr_lists(indptrs, indices, weights, staging=slot.staging_np, threads=self.pack_threads)
                return info
            except ValueError as e:
                if "staging buffer" not in str(e):
                    raise
                slot.ensure_staging(int(str(e).split("(")[1].split()[0]))

    def _batch_struct(self, slot: _Slot, info):
        base = slot.dev.data_ptr()
        return _lib.DgcnBatch(int(info.num_graphs), int(info.num_nodes), int(info.num_edges), int(info.max_nodes),
                              int(info.max_graph_edges), base + int(info.off_graph_ptr), base + int(info.off_row_ptr),
                              base + int(info.off_col_idx))

    def supported(self, slot: _Slot, info) -> bool:
        """Does ``dgcn_solve_batch`` (fused kernel or any-size path) take this (packed) batch with this model?"""
        return (int(info.max_degree) < self.table.numel() and
                int(self.lib.dgcn_solve_path(C.byref(self._batch_struct(slot, info)), C.byref(self.model.c))) != 0)

    def _launch(self, slot: _Slot, info) -> _Slot:
        """Device stage: one copy in, one fused launch, one copy out, all on the slot's stream; returns at once."""
        t = self.torch
        n, B, e_ = int(info.num_nodes), int(info.num_graphs), int(info.num_edges)
        if int(info.max_degree) >= self.table.numel():
            raise _lib.DgcnError("vertex degree %d beyond the pipeline's d^-1/2 table" % int(info.max_degree))
        slot.ensure_out(n, B, self.want_scores)
        total = int(info.total_bytes)
        base = slot.dev.data_ptr()
        bc = self._batch_struct(slot, info)
        if not self.lib.dgcn_solve_path(C.byref(bc), C.byref(self.model.c)):
            raise _lib.DgcnError("this model / batch shape is outside the fused kernel: use mwis_dqn_call.solve_host_batch")
        need = int(self.lib.dgcn_solve_workspace(C.byref(bc), C.byref(self.model.c)))
        if slot.ws is None or slot.ws.numel() < need:
            slot.ws = t.empty(max(need, 256), dtype=t.uint8, device=self.eng.device)
        ob = slot.out.data_ptr()
        lay = slot.layout
        with t.cuda.stream(slot.stream):
            slot.dev[:total].copy_(slot.staging[:total], non_blocking=True)
            _lib.check(self.lib.dgcn_solve_batch(
                C.byref(bc), C.byref(self.model.c), self.table.data_ptr(), int(self.table.numel()), None, self.x_const,
                (base + int(info.off_weights)) if int(info.off_weights) >= 0 else None, 1 if self.predict == "mwis" else 0,
                (ob + lay["scores"][0]) if self.want_scores else None, ob + lay["state"][0],
                ob + lay["rounds"][0], ob + lay["totals"][0], ob + lay["status"][0], slot.ws.data_ptr(), need,
                C.c_void_p(slot.stream.cuda_stream)), "dgcn_solve_batch")
            slot.res.copy_(slot.out, non_blocking=True)
            slot.event.record(slot.stream)
        slot.busy, slot.shape = True, (n, B)
        return slot

    def _next_slot(self) -> _Slot:
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        if slot.busy:
            raise RuntimeError("pipeline slot still holds an unread result: call result() before submitting %d more batches"
                               % len(self.slots))
        return slot

    def submit(self, indptrs: Sequence[np.ndarray], indices: Sequence[np.ndarray], weights: Sequence[np.ndarray]) -> _Slot:
        """Pack + enqueue one batch; returns its slot at once (``result(slot)`` waits for it)."""
        slot = self._next_slot()
        return self._launch(slot, self._pack(slot, indptrs, indices, weights))

    def result(self, slot: _Slot, copy: bool = True):
        """Wait for the slot's batch -> {"state", "totals", "rounds"} NumPy arrays in host memory (copies by default:
        the pinned buffer is re-used by the next batch in this slot)."""
        slot.event.synchronize()
        n, B = slot.shape
        host = slot.res_np
        out = {}
        for name, count in (("state", n), ("totals", B), ("rounds", B), ("status", 1)) + ((("scores", n),) if self.want_scores else ()):
            o, nb, dt = slot.layout[name]
            v = host[o:o + nb].view(_NP[dt])[:count]
            out[name] = v.copy() if copy else v
        slot.busy = False
        bits = int(out.pop("status")[0])
        if bits:
            with self.torch.cuda.stream(slot.stream):
                slot.out.zero_()
            Engine.check_status_bits(bits)
        return out

    def solve_many(self, batches, copy: bool = True):
        """Generator: feed an iterable of (indptrs, indices, weights) batches through the pipeline, yielding results in
        order.  While the GPU copies / solves batch k the host packs batch k+1 on a helper thread (the packer is native
        and releases the GIL) and this thread enqueues and collects: the stages overlap across batches."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        depth = len(self.slots)
        it = iter(batches)
        inflight = deque()
        if depth < 2:  # one slot: nothing to overlap with - pack, launch, collect, batch by batch
            for b in it:
                yield self.result(self.submit(*b), copy)
            return
        with ThreadPoolExecutor(max_workers=1) as ex:
            def start_pack():
                b = next(it, None)
                if b is None:
                    return None
                slot = self._next_slot()  # free by construction: its previous result has been yielded
                return slot, ex.submit(self._pack, slot, *b)
            nxt = start_pack()
            while nxt is not None:
                slot, fut = nxt
                info = fut.result()
                if len(inflight) == depth - 1:
                    yield self.result(inflight.popleft(), copy)  # frees the slot the next pack writes into
                nxt = start_pack()
                inflight.append(self._launch(slot, info))
            while inflight:
                yield self.result(inflight.popleft(), copy)


# Objects that are still open when the interpreter exits (the process-wide one-slot instances behind heuristics.* / solve_mwis
# are never closed by their users) are closed by an atexit handler - i.e. BEFORE the interpreter starts tearing modules down,
# while the HIP runtime, the streams and the pinned buffers they own are all still there - not by __del__ at some point of the
# finalisation.
import atexit
import weakref

_LIVE = weakref.WeakSet()


def _close_all():
    for hs in list(_LIVE):
        try:
            hs.close()
        except Exception:
            pass


atexit.register(_close_all)


class HostSolver:
    """``dgcn_host_solver_*`` (include/dgcn.h): pack -> copy in -> fused launch -> copy out in native code, ``depth``
    batches in flight.  ``submit`` returns a slot number at once; ``result(slot)`` waits for it."""

    def __init__(self, engine: Engine, model: Optional[DeviceModel], depth: int = 1, predict: str = "mwis",
                 pack_threads: int = 0, want_scores: bool = False):
        """``model=None``: no GCN, the plain local greedy search with the weights as priorities."""
        self.eng, self.model, self.lib = engine, model, engine.lib
        self.want_scores = want_scores and model is not None
        self.table = engine._dinv(4095)
        self.handle = C.c_void_p()
        x_const = float(np.float32(1.0 / model.in_dim)) if model is not None else 1.0
        with engine.torch.cuda.device(engine.device):
            _lib.check(self.lib.dgcn_host_solver_create(C.byref(model.c) if model is not None else None, self.table.data_ptr(),
                                                        int(self.table.numel()), 1 if predict == "mwis" else 0, x_const,
                                                        1 if self.want_scores else 0, int(depth), int(pack_threads),
                                                        C.byref(self.handle)),
                       "dgcn_host_solver_create")
        self.depth = int(depth)
        # the process-wide one-slot instances behind heuristics.* / solve_mwis are shared by every caller thread: a call and
        # the reading of its result (views into the slot's pinned memory) happen under this lock
        import threading
        self.lock = threading.RLock()
        self._p = [C.c_void_p() for _ in range(4)]
        self._i = [C.c_int32() for _ in range(3)]
        # one-call path of the CPython helper (csrc/pyptr.c): submit + result without the interpreter in between
        from .batch import _pyptr
        helper = _pyptr()
        self._solve_lists = getattr(helper, "solve_lists", None)
        self._fn = (C.cast(self.lib.dgcn_host_solver_submit, C.c_void_p).value, C.cast(self.lib.dgcn_host_solver_result, C.c_void_p).value)
        _LIVE.add(self)

    def close(self):
        if self.handle:
            self.lib.dgcn_host_solver_destroy(self.handle)
            self.handle = C.c_void_p()
        _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, indptrs: Sequence[np.ndarray], indices: Sequence[np.ndarray], weights: Optional[Sequence[np.ndarray]]) -> int:
        """Arrays must be contiguous, indptr / indices all int32 or all int64, weights float64 (TypeError otherwise)."""
        from .batch import _addresses
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
        # the last indptr entry of every graph must be the length of its indices array (the library cannot see lengths)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        info = _lib.DgcnPackInfo()
        nnz = np.empty(max(B, 1), dtype=np.int64)
        _lib.check(self.lib.dgcn_pack_measure(vp(ap), vp(nn), B, isz, 1 if weights is not None else 0, C.byref(info), vp(nnz)),
                   "dgcn_pack_measure")
        if B and not np.array_equal(ci[:B], nnz[:B]):
            raise ValueError("an indices array does not match its indptr")
        slot = self.lib.dgcn_host_solver_submit(self.handle, vp(ap), vp(ai), vp(aw) if aw is not None else None, vp(nn), B, isz)
        if slot < 0:
            _lib.check(slot, "dgcn_host_solver_submit")
        self._nn = nn
        return slot

    def result(self, slot: int, copy: bool = True):
        p, i = self._p, self._i
        _lib.check(self.lib.dgcn_host_solver_result(self.handle, slot, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3]),
                                                    C.byref(i[0]), C.byref(i[1]), C.byref(i[2])), "dgcn_host_solver_result")
        n, B = i[1].value, i[2].value
        Engine.check_status_bits(i[0].value)

        def view(ptr, count, ctype, dtype):
            if count == 0 or not ptr.value:
                return np.zeros(0, dtype)
            a = np.frombuffer((ctype * count).from_address(ptr.value), dtype=dtype)
            return a.copy() if copy else a
        out = {"state": view(p[0], n, C.c_uint8, np.uint8), "totals": view(p[1], B, C.c_double, np.float64),
               "rounds": view(p[2], B, C.c_int32, np.int32)}
        if self.want_scores:
            out["scores"] = view(p[3], n, C.c_float, np.float32)
        return out

    def solve(self, indptrs, indices, weights, copy: bool = True):
        if self._solve_lists is not None and copy and len(indptrs) <= 64:
            r = self._solve_lists(self._fn[0], self._fn[1], self.handle.value, indptrs, indices, weights)
            if isinstance(r, int):
                _lib.check(r, "dgcn_host_solver_submit")
            state, totals, rounds, scores, bits, nn = r
            Engine.check_status_bits(bits)
            self._nn = np.frombuffer(nn, np.int32)
            out = {"state": np.frombuffer(state, np.uint8), "totals": np.frombuffer(totals, np.float64),
                   "rounds": np.frombuffer(rounds, np.int32)}
            if self.want_scores:
                out["scores"] = np.frombuffer(scores, np.float32) if scores is not None else np.zeros(0, np.float32)
            return out
        return self.result(self.submit(indptrs, indices, weights), copy)

    def solve_many(self, batches, copy: bool = True, depth: Optional[int] = None):
        """Generator: an iterable of (indptrs, indices, weights) batches -> results in order, up to ``depth`` (default: all
        slots) batches in flight.  ``submit`` packs on the library's worker threads and returns as soon as the copies and
        the launch are queued, so batch k + 1 is packed while the GPU works on batch k."""
        from collections import deque
        inflight = deque()
        depth = int(depth or self.depth)
        for b in batches:
            if len(inflight) == depth:
                yield self.result(inflight.popleft(), copy)
            inflight.append(self.submit(*b))
        while inflight:
            yield self.result(inflight.popleft(), copy)
