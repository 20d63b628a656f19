#!/usr/bin/env python3
"""What a k_fused launch costs when every graph has a CU to itself and almost no vertices (the floor a late step of a
search pays): 64 graphs of n vertices, 20 layers, 512- and 1 024-thread workgroups, with the -DDGCN_DIAG build's
ablation switches.   DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/solo_floor.py [graphs=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd import _lib
from distgcn_amd.engine import Engine, DeviceModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("cuda:0")
model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
rows = [("everything", 0), ("no weight fetch", 8), ("no aggregation", 1), ("no transform", 2), ("neither", 3), ("neither, no weight fetch", 11),
        ("layers only: neither + no greedy", 7)]
for n in (16, 64, 128, 256, 500):
    hb = datagen.er_batch(B, n, min(0.9, 10.0 / n))
    db = eng.upload(hb)
    out = eng.solve_buffers(db, True)
    for block in ("512", "1024"):
        _lib.set_option("fused_block", int(block))
        line = []
        for name, bits in rows:
            _lib.set_option("diag_flags", bits)
            for _ in range(30):
                eng.solve_fused(db, model, out=out)
            torch.cuda.synchronize()
            eng.timing(True)
            for _ in range(200):
                eng.solve_fused(db, model, out=out)
            torch.cuda.synchronize(); eng.timing(False)
            ms, cnt = eng.timing_read("fused_solve")
            line.append("%s %.1f" % (name, ms / cnt * 1e3))
        print("n=%d block=%s (us per launch): %s" % (n, block, "; ".join(line)), flush=True)
