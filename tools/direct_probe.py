#!/usr/bin/env python3
"""Host-solver call time for small batches: packed batch read in place from pinned memory vs copied first.
python tools/direct_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distgcn_amd import datagen
from distgcn_amd import _lib
from distgcn_amd.engine import Engine, DeviceModel
from distgcn_amd.serving import HostSolver
eng = Engine("cuda:0"); dm = DeviceModel(datagen.random_model(20, 32), "cuda:0")
hs = HostSolver(eng, dm, depth=1)
for B in (1, 4, 8, 16, 32, 63):
    hb = datagen.er_batch(B, 200, 0.1)
    ps, cs, ws = [], [], []
    for n0, n1 in hb.graph_slices():
        e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
        ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0, dtype=np.int32)); cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0, dtype=np.int32))
        ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
    line = "B=%2d (%4d KB packed):" % (B, (hb.num_edges * 4 + hb.num_nodes * 12) // 1024)
    for name, val in (("copied", "0"), ("in place", str(64 << 20))):
        _lib.set_option("host_direct_bytes", int(val))
        for _ in range(30): hs.solve(ps, cs, ws)
        t0 = time.perf_counter()
        for _ in range(300): hs.solve(ps, cs, ws)
        line += "  %s %6.1f us" % (name, (time.perf_counter() - t0) / 300 * 1e6)
    print(line)
