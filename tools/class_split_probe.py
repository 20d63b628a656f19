#!/usr/bin/env python3
"""Would launching a mixed batch as size classes on concurrent streams pay?  The C4 share (500 BA test2-mix graphs, l = 20)
as ONE launch against the same graphs split by image size into classes, one launch per class, each on its own stream.
   python tools/class_split_probe.py [bounds in KB of LDS image, default "80"] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.batch import HostBatch
from distgcn_amd.engine import Engine, DeviceModel

dev = "cuda:0"
hb = datagen.ba_test2_batch(500)
layers = datagen.random_model(20, 32)
gp, rp = hb.graph_ptr, hb.row_ptr
n = np.diff(gp); e = rp[gp[1:]] - rp[gp[:-1]]
img = (n * 256 + (e + n + 34) * 6 + n * 8 + 256) / 1024.0


def sub(ids):
    ps, cs, ws = [], [], []
    for g in ids:
        a, b = gp[g], gp[g + 1]
        r = rp[a:b + 1]
        ps.append((r - r[0]).astype(np.int32)); cs.append((hb.col_idx[r[0]:r[-1]] - a).astype(np.int32)); ws.append(hb.weights[a:b])
    return HostBatch.from_csr_lists(ps, cs, ws)


def timed(launch, steps=300):
    for _ in range(50): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


eng = Engine(dev); model = DeviceModel(layers, dev)
db = eng.upload(hb); out = eng.solve_buffers(db, False)
print("one launch, 500 graphs: %.1f us per step" % timed(lambda: eng.solve_fused(db, model, out=out, want_scores=False)), flush=True)
for spec in (sys.argv[1:] or ["80", "53,80", "80,110", "53,80,110"]):
    bounds = [float(x) for x in spec.split(",")] + [1e9]
    classes, lo = [], 0.0
    for hi in bounds:
        ids = np.nonzero((img > lo) & (img <= hi))[0]
        lo = hi
        if len(ids): classes.append(ids)
    classes = classes[::-1]  # the large class first
    engs = [Engine(dev) for _ in classes]
    dbs = [en.upload(sub(ids)) for en, ids in zip(engs, classes)]
    outs = [en.solve_buffers(d, False) for en, d in zip(engs, dbs)]
    streams = [torch.cuda.Stream() for _ in classes]
    main = torch.cuda.current_stream()

    def launch():
        ev0 = torch.cuda.Event(); ev0.record(main)
        for en, d, o, s in zip(engs, dbs, outs, streams):
            s.wait_event(ev0)
            with torch.cuda.stream(s):
                en.solve_fused(d, model, out=o, want_scores=False)
            ev = torch.cuda.Event(); ev.record(s); main.wait_event(ev)

    alone = []
    for en, d, o in zip(engs, dbs, outs):
        alone.append(timed(lambda: en.solve_fused(d, model, out=o, want_scores=False)))
    print("classes by image KB <= %s: sizes %s, each alone %s us; concurrently (joined per step): %.1f us per step"
          % (spec, [len(c) for c in classes], ["%.1f" % a for a in alone], timed(launch)), flush=True)
