#!/usr/bin/env python3
"""What the phases of a k_fused C3 launch cost: the -DDGCN_DIAG build (tools/build_diag.sh) with the hidden layers'
aggregation and / or transform switched off, two workgroups per CU and one.  Results of the ablated launches are
meaningless; only their duration is read.
   DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/ablate_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
hb = datagen.er_batch(B, 200, 0.1)
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
out = eng.solve_buffers(db, True)
for _ in range(200):
    eng.solve_fused(db, model, out=out)
torch.cuda.synchronize()
rows = [("everything", 0), ("no aggregation (hidden layers)", 1), ("no transform (hidden layers)", 2), ("neither", 3),
        ("neither, no weight fetch", 11), ("no greedy rounds", 4), ("layers only: neither + no greedy", 7),
        ("neighbour parities forced conflict-free", 16), ("... and no transform", 18)]
# (until round 5 the diag build could also pad the LDS to force one workgroup per CU: profiles/r03_fused_ablation.txt has that table)
if True:
    for name, bits in rows:
        _lib.set_option("diag_flags", bits)
        for _ in range(20):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(100):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(False)
        ms, n = eng.timing_read("fused_solve")
        print("  %-40s %8.1f us per launch" % (name, ms / n * 1e3))
