#!/usr/bin/env python3
"""A/B of k_fused launch options (dgcn_set_option keys) in ONE process, interleaved rounds (cdna guide rule 24):
   python tools/ab_fused.py "fused_block=512" "fused_block=1024" ...   (each argument is one variant: key=value,key=value; "" = defaults)
   DGCN_AB_GRAPHS / DGCN_AB_KIND=er|ba|ermix pick the batch (default 500 ER(200, 0.1))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED

variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in (sys.argv[1:] or [""])]
keys = sorted({k for v in variants for k in v})
nB = int(os.environ.get("DGCN_AB_GRAPHS", "500"))
kind = os.environ.get("DGCN_AB_KIND", "er")
if kind == "ermix":  # ER(n, 0.1) with n cycling through 80 .. 200: a mixed batch whose images all leave two workgroups per CU
    from distgcn_amd.batch import HostBatch
    ps, cs, ws = [], [], []
    for g in range(nB):
        rng = np.random.default_rng(datagen.SEED0 + 2_000_000 + g)
        ip, ix = datagen.er_graph((80, 120, 160, 200, 100, 140, 180)[g % 7], 0.1, rng)
        ps.append(ip); cs.append(ix); ws.append(rng.random(len(ip) - 1))
    hb = HostBatch.from_csr_lists(ps, cs, ws)
else:
    hb = datagen.ba_test2_batch(nB) if kind == "ba" else datagen.er_batch(nB, 200, 0.1)
defaults = _lib.option_defaults()
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
out = eng.solve_buffers(db, True)
ref = None
res = {i: [] for i in range(len(variants))}
for _ in range(300):
    eng.solve_fused(db, model, out=out)
torch.cuda.synchronize()
for rnd in range(7):
    for i, v in enumerate(variants):
        for k in keys:
            _lib.set_option(k, defaults[k])
        for k, val in v.items():
            _lib.set_option(k, int(val))
        for _ in range(20):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(100):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(False)
        ms, n = eng.timing_read("fused_solve")
        res[i].append(ms / n * 1e3)
        sc = out["scores"].cpu().numpy().copy(); st = out["state"].cpu().numpy().copy()
        if ref is None:
            ref = (sc, st)
        assert np.array_equal(sc.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(st, ref[1]), "variant %s changes the results" % v
for i, v in enumerate(variants):
    print("%-60s median %7.2f us  min %7.2f us" % (v, float(np.median(res[i])), min(res[i])))
