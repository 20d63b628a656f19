#!/usr/bin/env python3
"""Time k_fused on the C3 batch under the environment it is started with (no result check): experiments only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
hb = datagen.er_batch(500, 200, 0.1)
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
out = eng.solve_buffers(db, True)
for _ in range(300):
    eng.solve_fused(db, model, out=out)
torch.cuda.synchronize()
res = []
for rnd in range(5):
    eng.timing(True)
    for _ in range(100):
        eng.solve_fused(db, model, out=out)
    torch.cuda.synchronize(); eng.timing(False)
    ms, n = eng.timing_read("fused_solve")
    res.append(ms / n * 1e3)
print("rounds per graph: mean %.1f max %d" % (out["rounds"].float().mean().item(), out["rounds"].max().item()))
print("%s median %7.2f us  min %7.2f us" % (" ".join(sys.argv[1:]), float(np.median(res)), min(res)))
