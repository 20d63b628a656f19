"""Host-to-host serving loop: CSR lists in host memory -> membership / totals / rounds in host memory.

The reference's evaluation loop handles one graph at a time: load, ``makestate``, ``sess.run``, greedy, ratio
(``mwis_dqn_test.py:304-321``).  Here a batch goes through four stages that overlap across batches:

    pack (host threads, ``dgcn_pack_batch`` straight into pinned memory)
      -> one host-to-device copy -> ONE fused launch (``dgcn_solve_batch``) -> one device-to-host copy

``SolvePipeline`` keeps ``depth`` slots (pinned staging, device buffers, a stream and an event each); while the
GPU works on slot k the host packs slot k+1.  Only shapes the fused kernel takes are served here
(``Engine.solve_supported``); other shapes go through ``mwis_dqn_call.solve_host_batch``.
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
        """Does the fused kernel take this (packed) batch with this model?"""
        return (int(info.max_degree) < self.table.numel() and
                bool(self.lib.dgcn_solve_supported(C.byref(self._batch_struct(slot, info)), C.byref(self.model.c))))

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
        if not self.lib.dgcn_solve_supported(C.byref(bc), C.byref(self.model.c)):
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
                base + int(info.off_weights), 1 if self.predict == "mwis" else 0,
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
