#!/usr/bin/env python3
"""Host packing time of one C3 batch (500 ER N=200 graphs as separate int32 CSR arrays): ordinary block-diagonal format
(dgcn_pack_batch) against the compact transfer format (dgcn_pack_compact_batch).  CPU only.  python tools/pack_probe.py [threads]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distgcn_amd import _lib, datagen
from distgcn_amd.batch import _addresses
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
hb = datagen.er_batch(500, 200, 0.1)
ps, cs, ws = [], [], []
for n0, n1 in hb.graph_slices():
    e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
    ps.append((hb.row_ptr[n0:n1 + 1] - e0).astype(np.int32)); cs.append((hb.col_idx[e0:e1] - n0).astype(np.int32)); ws.append(hb.weights[n0:n1].copy())
B = len(ps)
ap, cp, isz = _addresses(ps, 0); ai, _, _ = _addresses(cs, isz); aw, _, _ = _addresses(ws, 64)
nn = (cp[:B] - 1).astype(np.int32)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
info, ci = _lib.DgcnPackInfo(), _lib.DgcnCompactInfo()
lib.dgcn_pack_measure(vp(ap), vp(nn), B, isz, 1, C.byref(info), None)
lib.dgcn_pack_compact_layout(C.byref(info), C.byref(ci))
buf = np.zeros(int(info.total_bytes), np.uint8)
def std(): return lib.dgcn_pack_batch(vp(ap), vp(ai), vp(aw), vp(nn), B, isz, vp(buf), buf.nbytes, C.byref(info), threads)
def cmp(): return lib.dgcn_pack_compact_batch(vp(ap), vp(ai), vp(aw), vp(nn), B, isz, vp(buf), buf.nbytes, C.byref(info), C.byref(ci), threads)
for name, fn in (("ordinary", std), ("compact", cmp), ("ordinary", std), ("compact", cmp)):
    assert fn() == 0
    t0 = time.perf_counter()
    for _ in range(300): fn()
    print("%-9s %.3f ms per batch (%d threads; %.1f MB written)" % (name, (time.perf_counter() - t0) / 300 * 1e3, threads,
          (int(info.total_bytes) if name == "ordinary" else int(ci.total_bytes)) / 1e6))
