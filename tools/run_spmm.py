#!/usr/bin/env python3
"""Run the batched SpMM kernel alone (for rocprofv3 --pmc passes): python tools/run_spmm.py [er|ba] [iters] [sets]
sets > 1: that many DISTINCT 500-graph batches visited round-robin, so that > 256 MiB of other traffic lies
between two uses of a line (out of the Infinity Cache: the HBM measurement).  sets < 0: ONE batch of |sets| * 500
graphs per launch (-8 = 4 000 graphs, 444 MB per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = Engine("cuda:0")
sets = []
count = 500
if nsets < 0:
    count, nsets = -nsets * 500, 1
for i in range(nsets):
    hb = datagen.er_batch(count, 200, 0.1, first_index=i * count) if kind == "er" else datagen.ba_test2_batch(count, first_index=i * count)
    db = eng.upload(hb)
    sets.append((hb, db, eng.supports(db), torch.randn(hb.num_nodes, 64, device="cuda"), torch.empty(hb.num_nodes, 32, device="cuda")))
for _ in range(iters):
    for hb, db, lap, Z, out in sets:
        eng.spmm(lap, Z[:, 32:], 32, ldz=64, graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs, max_nodes=hb.max_nodes,
                 Y0=Z, ldy0=64, act="leaky_relu", out=out)
torch.cuda.synchronize()
