#!/usr/bin/env python3
"""Mixed-size batches (C4: BA test2 mix): one launch vs residency classes on separate streams.
python tools/run_buckets.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hb = datagen.ba_test2_batch(500)
eng = Engine("cuda:0")
model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
sizes = np.diff(hb.graph_ptr).astype(np.int64)
nnz = (hb.row_ptr[hb.graph_ptr[1:]] - hb.row_ptr[hb.graph_ptr[:-1]]).astype(np.int64)
need = np.maximum(sizes, 64) * 32 * 4 * 2 + (nnz + 2 * sizes + 6) * 6 + sizes * 6 + 64
res = np.minimum(160 * 1024 // need, 4)
print("residency classes:", {int(r): int((res == r).sum()) for r in np.unique(res)})

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6

db = eng.upload(hb); out = eng.solve_buffers(db)
print("one launch: %.1f us" % timeit(lambda: eng.solve_fused(db, model, out=out)))
for scheme, classes in (("2 classes", [res >= 2, res < 2]), ("by residency", [res == r for r in sorted(np.unique(res), reverse=True)]),
                        ("by residency, big first", [res == r for r in sorted(np.unique(res))])):
    subs = [eng.upload(hb.select(np.flatnonzero(c))) for c in classes if c.any()]
    outs = [eng.solve_buffers(s) for s in subs]
    def seq():
        for s, o in zip(subs, outs): eng.solve_fused(s, model, out=o)
    print("%s, one stream: %.1f us" % (scheme, timeit(seq)))
    streams = [torch.cuda.Stream() for _ in subs]
    engs = [Engine("cuda:0") for _ in subs]  # one workspace per stream
    def par():
        cur = torch.cuda.current_stream()
        for s, o, st, e in zip(subs, outs, streams, engs):
            st.wait_stream(cur)
            with torch.cuda.stream(st): e.solve_fused(s, model, out=o)
        for st in streams: cur.wait_stream(st)
    print("%s, %d streams: %.1f us" % (scheme, len(subs), timeit(par)))
