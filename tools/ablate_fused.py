#!/usr/bin/env python3
"""What the phases of a k_fused C3 launch cost: the -DDGCN_DIAG build (tools/build_diag.sh) with the hidden layers'
aggregation and / or transform switched off, two workgroups per CU and one.  Results of the ablated launches are
meaningless; only their duration is read.
   DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/ablate_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
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
for pad in ("0", "40000"):
    os.environ["DGCN_FUSED_LDS_PAD"] = pad
    print("workgroups per CU: %s" % ("2" if pad == "0" else "1 (LDS padded)"))
    for name, bits in rows:
        os.environ["DGCN_FUSED_DIAG"] = str(bits)
        for _ in range(20):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(100):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(False)
        ms, n = eng.timing_read("fused_solve")
        print("  %-40s %8.1f us per launch" % (name, ms / n * 1e3))
