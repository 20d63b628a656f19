#!/usr/bin/env python3
"""C3 steps issued on one stream vs alternating on two (two engines = two workspaces): does the next launch's start hide
the previous one's tail?  python tools/time_two_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel
hb = datagen.er_batch(500, 200, 0.1)
engs = [Engine("cuda:0"), Engine("cuda:0")]
dbs = [e.upload(hb) for e in engs]
models = [DeviceModel(datagen.random_model(20, 32), "cuda:0") for _ in engs]
outs = [[e.solve_buffers(d, False) for _ in range(2)] for e, d in zip(engs, dbs)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(nstreams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            engs[k].solve_fused(dbs[k], models[k], out=outs[k][(i // nstreams) % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
for n in (1, 2): run(n, 300)
for rep in range(3):
    for n in (1, 2):
        print("%d stream(s): %.1f us per step" % (n, run(n, 1500)))
ref = outs[0][0]["state"].cpu().numpy(); got = outs[1][0]["state"].cpu().numpy()
print("same sets:", np.array_equal(ref, got))
