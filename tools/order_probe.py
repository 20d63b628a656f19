#!/usr/bin/env python3
"""C4 (BA test2 mix, 500 graphs): does the order in which graphs are dealt to workgroups matter?  The same graphs as they
come, largest first, smallest first.  python tools/order_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
hb = datagen.ba_test2_batch(500)
eng = Engine("cuda:0")
sizes = np.diff(hb.graph_ptr).astype(np.int64)
nnz = (hb.row_ptr[hb.graph_ptr[1:]] - hb.row_ptr[hb.graph_ptr[:-1]]).astype(np.int64)
print("N: min %d mean %.0f max %d;  entries: min %d mean %.0f max %d" % (sizes.min(), sizes.mean(), sizes.max(), nnz.min(), nnz.mean(), nnz.max()))
for layers in (20, 1):
    model = DeviceModel(datagen.random_model(layers, 32), "cuda:0")
    for name, ids in (("as they come", np.arange(500)), ("largest first", np.argsort(-(nnz + 8 * sizes), kind="stable")),
                      ("smallest first", np.argsort(nnz + 8 * sizes, kind="stable"))):
        db = eng.upload(hb.select(ids)); out = eng.solve_buffers(db)
        for _ in range(20): eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(True)
        for _ in range(100): eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(False)
        ms, n = eng.timing_read("fused_solve")
        print("l=%2d  %-15s %7.1f us per launch" % (layers, name, ms / n * 1e3))
