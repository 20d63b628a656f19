#!/usr/bin/env python3
"""Where a native host-to-host batch spends its time: submit() vs result() (C3 batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
from distgcn_amd.serving import HostSolver
hb = datagen.er_batch(500, 200, 0.1)
ps, cs, ws = [], [], []
for n0, n1 in hb.graph_slices():
    e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
    ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0, dtype=np.int32)); cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0, dtype=np.int32))
    ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
eng = Engine("cuda:0"); dm = DeviceModel(datagen.random_model(20, 32), "cuda:0")
hs = HostSolver(eng, dm, depth=3, pack_threads=8)
for _ in range(5): hs.solve(ps, cs, ws)
ts, tr = [], []
slots = []
for k in range(60):
    t0 = time.perf_counter(); slots.append(hs.submit(ps, cs, ws)); t1 = time.perf_counter(); ts.append(t1 - t0)
    if len(slots) == 3:
        t0 = time.perf_counter(); hs.result(slots.pop(0), copy=False); tr.append(time.perf_counter() - t0)
print("submit %.3f ms (min %.3f)  result %.3f ms (min %.3f)" % (np.median(ts) * 1e3, min(ts) * 1e3, np.median(tr) * 1e3, min(tr) * 1e3))
t0 = time.perf_counter(); s = hs.submit(ps, cs, ws); t1 = time.perf_counter(); hs.result(s); t2 = time.perf_counter()
print("alone: submit %.3f ms, result %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
