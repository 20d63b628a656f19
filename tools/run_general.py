#!/usr/bin/env python3
"""Run the any-size solve path alone (for rocprofv3 passes): python tools/run_general.py er500|mc900|erNxP [iters] [layers] [graphs]
(dgcn_solve_batch on graphs beyond the fused kernel: k_supports + k_big + k_lgs per call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
kind = sys.argv[1] if len(sys.argv) > 1 else "er500"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
if kind == "mc900":
    hb = datagen.multichannel_batch(B, 300, 0.03)
elif kind == "mc1500":
    hb = datagen.multichannel_batch(B, 500, 0.03)
elif kind == "er500":
    hb = datagen.er_batch(B, 500, 0.1)
else:
    n, p = kind[2:].split("x")
    hb = datagen.er_batch(B, int(n), float(p))
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, 32), "cuda:0")
print("path", eng.solve_path(db, model))
out = eng.solve_buffers(db, False)
eng.timing(True)
for _ in range(iters):
    res = eng.solve_fused(db, model, want_scores=False, out=out)
torch.cuda.synchronize(); eng.timing(False)
for fam in ("supports", "big_forward", "big_solve", "lgs", "fused_solve", "wide_solve", "transform", "spmm", "layer"):
    ms, n = eng.timing_read(fam)
    if n:
        print("%s %s l=%d: %.1f us avg over %d launches" % (fam, kind, nl, ms / n * 1e3, n))
