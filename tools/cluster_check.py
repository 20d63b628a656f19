#!/usr/bin/env python3
"""Cluster variant of k_fused (one graph on K workgroups) against the ordinary launch: same bits, launch time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
eng = Engine("cuda:0")
for layers in ((20,) if len(sys.argv) > 1 else (20, 3, 2)):
    model = DeviceModel(datagen.random_model(layers, 32), "cuda:0")
    for B, n in ((1, 200), (3, 200), (8, 200), (16, 200), (20, 120), (32, 200), (64, 200), (5, 500), (2, 77), (4, 300), (8, 150)):
        hb = datagen.er_batch(B, n, 0.1 if n <= 200 else 0.02)
        db = eng.upload(hb)
        res = {}
        for mode in ("0", "auto"):
            # (the switch is an atomic inside the library, read from the environment once: set it through the C ABI)
            eng.lib.dgcn_set_cluster(0 if mode == "0" else (int(sys.argv[1]) if len(sys.argv) > 1 else -1))
            out = eng.solve_buffers(db, True)
            for _ in range(20): eng.solve_fused(db, model, out=out, want_scores=True)
            torch.cuda.synchronize()
            eng.timing(True)
            for _ in range(100): eng.solve_fused(db, model, out=out, want_scores=True)
            torch.cuda.synchronize(); eng.timing(False)
            ms, cnt = eng.timing_read("fused_solve")
            res[mode] = (ms / cnt * 1e3, {k: out[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")})
        a, b = res["0"][1], res["auto"][1]
        same = all(np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)) for k in a)
        print("layers %2d  B=%3d N=%3d  one workgroup per graph %7.1f us   cluster %7.1f us   identical: %s  status %d" %
              (layers, B, n, res["0"][0], res["auto"][0], same, int(b["status"][0])))
