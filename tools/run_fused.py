#!/usr/bin/env python3
"""Run the fused solve kernel alone (for rocprofv3 passes): python tools/run_fused.py [er|ba] [iters] [layers] [graphs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B = int(sys.argv[4]) if len(sys.argv) > 4 else 500
hb = (datagen.er_batch(B, int(kind[2:] or 200), 0.1) if kind.startswith("er") else datagen.ba_test2_batch(B))
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, 32), "cuda:0")
eng.timing(True)
for _ in range(iters):
    res = eng.solve(db, model, mode=MODE_FUSED)
torch.cuda.synchronize(); eng.timing(False)
ms, n = eng.timing_read("fused_solve")
print("fused_solve %s l=%d: %.1f us avg over %d launches" % (kind, nl, ms / n * 1e3, n))
