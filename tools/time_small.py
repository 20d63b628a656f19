#!/usr/bin/env python3
"""k_fused launch time for small batches (one graph has a CU to itself): python tools/time_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
eng = Engine("cuda:0"); model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
for B in ((1,) if len(sys.argv) > 1 else (1, 8, 64, 250, 256, 500, 1000, 4000)):
    hb = datagen.er_batch(B, 200, 0.1)
    db = eng.upload(hb); out = eng.solve_buffers(db, True)
    for _ in range(50): eng.solve_fused(db, model, out=out)
    torch.cuda.synchronize()
    eng.timing(True)
    for _ in range(200 if B <= 500 else 40): eng.solve_fused(db, model, out=out)
    torch.cuda.synchronize(); eng.timing(False)
    ms, n = eng.timing_read("fused_solve")
    print("B=%5d  %8.1f us per launch  %8.3f M graphs/s" % (B, ms / n * 1e3, B / (ms / n * 1e3)))
