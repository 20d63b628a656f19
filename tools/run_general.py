#!/usr/bin/env python3
"""Run the any-size solve path alone (for rocprofv3 passes): python tools/run_general.py er500|mc900|erNxP [iters] [layers] [graphs] [hidden] [supports]
(dgcn_solve_batch on graphs beyond the fused kernel: one launch of k_big / k_big2 / k_wide1 where they take the shape; hidden < 32:
the zero-padded copy on the same kernels; supports = 3: [I, L, L.L] through the layer-by-layer kernels.  DGCN_OPTIONS="big=0,big2=0"
in the environment: the layer-by-layer chain for comparison)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
kind = sys.argv[1] if len(sys.argv) > 1 else "er500"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
hidden = int(sys.argv[5]) if len(sys.argv) > 5 else 32
nsup = int(sys.argv[6]) if len(sys.argv) > 6 else 2
if kind == "mc900":
    hb = datagen.multichannel_batch(B, 300, 0.03)
elif kind == "mc1500":
    hb = datagen.multichannel_batch(B, 500, 0.03)
elif kind == "er500":
    hb = datagen.er_batch(B, 500, 0.1)
else:
    n, p = kind[2:].split("x")
    hb = datagen.er_batch(B, int(n), float(p))
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, hidden, num_supports=nsup), "cuda:0")
print("path", eng.solve_path(db, model))
out = eng.solve_buffers(db, False)
eng.timing(True)
for _ in range(iters):
    res = eng.solve_fused(db, model, want_scores=False, out=out)
torch.cuda.synchronize(); eng.timing(False)
tot = 0.0
for fam in ("fused_pad", "supports", "supports2_count", "supports2_scan", "supports2_fill", "big_forward", "big_solve", "lgs", "fused_solve", "wide_solve", "transform", "spmm", "layer"):
    ms, n = eng.timing_read(fam)
    if n:
        tot += ms / iters
        print("%s %s l=%d c=%d: %.1f us avg over %d launches" % (fam, kind, nl, hidden, ms / n * 1e3, n))
print("kernels per call %s l=%d c=%d supports=%d: %.1f us" % (kind, nl, hidden, nsup, tot * 1e3))
