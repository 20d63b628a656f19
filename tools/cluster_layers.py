#!/usr/bin/env python3
"""From how many layers on does the cluster variant pay?  One N = 200 graph, forced on vs off.  python tools/cluster_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
eng = Engine("cuda:0")
for B, n in ((1, 200), (16, 200), (4, 300), (8, 128)):
    hb = datagen.er_batch(B, n, 0.1 if n <= 200 else 0.05); db = eng.upload(hb)
    for layers in (3, 4, 5, 6, 7, 8, 10, 12):
        model = DeviceModel(datagen.random_model(layers, 32), "cuda:0")
        r = {}
        for mode in ("0", "8"):
            eng.lib.dgcn_set_cluster(int(mode))
            out = eng.solve_buffers(db, False)
            for _ in range(30): eng.solve_fused(db, model, out=out)
            torch.cuda.synchronize(); eng.timing(True)
            for _ in range(150): eng.solve_fused(db, model, out=out)
            torch.cuda.synchronize(); eng.timing(False)
            ms, cnt = eng.timing_read("fused_solve"); r[mode] = ms / cnt * 1e3
        print("B=%2d N=%3d layers %2d: one workgroup %6.1f us, cluster %6.1f us" % (B, n, layers, r["0"], r["8"]))
