#!/usr/bin/env python3
"""Run the batched SpMM kernel alone (for rocprofv3 --pmc passes): python tools/run_spmm.py [er|ba] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hb = datagen.er_batch(500, 200, 0.1) if kind == "er" else datagen.ba_test2_batch(500)
eng = Engine("cuda:0"); db = eng.upload(hb); lap = eng.supports(db)
Z = torch.randn(hb.num_nodes, 64, device="cuda"); out = torch.empty(hb.num_nodes, 32, device="cuda")
for _ in range(iters):
    eng.spmm(lap, Z[:, 32:], 32, ldz=64, graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs, max_nodes=hb.max_nodes,
             Y0=Z, ldy0=64, act="leaky_relu", out=out)
torch.cuda.synchronize()
