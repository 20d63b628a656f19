#!/usr/bin/env python3
"""Host-to-host pipeline rate vs packer threads and pipeline depth (C3 batch): python tools/e2e_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.batch import pack_csr_lists
from distgcn_amd.engine import Engine, DeviceModel
from distgcn_amd.serving import SolvePipeline
hb = datagen.er_batch(500, 200, 0.1)
ps, cs, ws = [], [], []
for n0, n1 in hb.graph_slices():
    e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
    ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0, dtype=np.int32)); cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0, dtype=np.int32))
    ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
eng = Engine("cuda:0"); dm = DeviceModel(datagen.random_model(20, 32), "cuda:0")
staging = np.empty(64 << 20, np.uint8)
for th in (1, 2, 4, 8, 16, 32, 64):
    pack_csr_lists(ps, cs, ws, staging=staging, threads=th)
    t = time.perf_counter()
    for _ in range(20): pack_csr_lists(ps, cs, ws, staging=staging, threads=th)
    print("pack threads %2d: %.3f ms per batch" % (th, (time.perf_counter() - t) / 20 * 1e3))
for depth in (2, 3, 4):
    for th in (4, 8, 16, 32):
        pipe = SolvePipeline(eng, dm, depth=depth, pack_threads=th)
        for _ in pipe.solve_many(((ps, cs, ws) for _ in range(20)), copy=False): pass
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in pipe.solve_many(((ps, cs, ws) for _ in range(300)), copy=False): pass
        dt = time.perf_counter() - t
        print("depth %d pack threads %2d: %.3f ms per batch, %.2f M graphs/s" % (depth, th, dt / 300 * 1e3, 500 * 300 / dt / 1e6))

from distgcn_amd.serving import HostSolver
for depth in (2, 3, 4):
    for th in (4, 8, 16):
        hs = HostSolver(eng, dm, depth=depth, pack_threads=th)
        for _ in hs.solve_many(((ps, cs, ws) for _ in range(20)), copy=False): pass
        t = time.perf_counter()
        for _ in hs.solve_many(((ps, cs, ws) for _ in range(300)), copy=False): pass
        dt = time.perf_counter() - t
        print("native depth %d pack threads %2d: %.3f ms per batch, %.2f M graphs/s" % (depth, th, dt / 300 * 1e3, 500 * 300 / dt / 1e6))
        hs.close()
