#!/usr/bin/env python3
"""Where a mixed batch's k_fused launch spends its time, workgroup by workgroup (needs the -DDGCN_DIAG build: DGCN_LIB=.../libdgcn_diag.so):
start / duration / end of every workgroup in dispatch order, from wave 0's phase clocks.   tools/timeline_fused.py ba 20 500"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
kind = sys.argv[1] if len(sys.argv) > 1 else "ba"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nb_graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 500
hb = (datagen.er_batch(nb_graphs, int(kind[2:] or 200), 0.1) if kind.startswith("er") else datagen.ba_test2_batch(nb_graphs))
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, 32), "cuda:0")
for _ in range(3): eng.solve(db, model, mode=MODE_FUSED)
torch.cuda.synchronize()
st = torch.zeros(hb.num_graphs * 64, dtype=torch.int64, device="cuda")
_lib.set_option("diag_stamps", st.data_ptr())
eng.solve(db, model, mode=MODE_FUSED); torch.cuda.synchronize()
_lib.set_option("diag_stamps", 0)
raw = st.cpu().numpy().reshape(-1, 64)
us = raw.astype(np.float64) / 2400.0
start = raw[:, 14].astype(np.float64) / 100.0     # s_memrealtime (100 MHz) at the workgroup's first instruction / behind its last store
end = raw[:, 15].astype(np.float64) / 100.0
t0 = start.min()
start -= t0; end -= t0
dur = end - start
cu = (raw[:, 13].astype(np.int64) << 8) | ((raw[:, 12].astype(np.int64) >> 8) & 0xff)   # (XCC, SE / SH / CU of HW_ID)
sl = hb.graph_slices()
nv = np.array([b - a for a, b in sl]); ne = np.array([int(hb.row_ptr[b] - hb.row_ptr[a]) for a, b in sl])
order = np.argsort(start, kind="stable")
print("graphs %d; launch span by the stamps %.1f us; sum of durations / 256 = %.1f us; longest workgroup %.1f us" % (hb.num_graphs, end.max(), dur.sum() / 256.0, dur.max()))
print("workgroups that start within 5 us of the launch: %d; later starters: %d" % (int((start < 5).sum()), int((start >= 5).sum())))
late = order[(start[order] >= 5)]
if late.size:
    print("later starters: start min %.1f median %.1f max %.1f; their durations min %.1f median %.1f max %.1f; ends max %.1f" % (
        start[late].min(), np.median(start[late]), start[late].max(), dur[late].min(), np.median(dur[late]), dur[late].max(), end[late].max()))
first = order[(start[order] < 5)]
print("first round: durations min %.1f median %.1f max %.1f" % (dur[first].min(), np.median(dur[first]), dur[first].max()))
# duration as a function of size: least squares dur ~ c0 + a * vertices + b * entries, by round
for name, idx in (("first round", first), ("later starters", late)):
    if idx.size > 10:
        A = np.stack([np.ones(idx.size), nv[idx], ne[idx]], axis=1)
        c, *_ = np.linalg.lstsq(A, dur[idx], rcond=None)
        res = dur[idx] - A @ c
        print("%s: duration ~ %.1f + %.3f x vertices + %.5f x entries  (rms residual %.1f us; vertex weight in entries %.0f)" % (name, c[0], c[1], c[2], float(np.sqrt((res ** 2).mean())), c[1] / max(c[2], 1e-9)))
print("ends: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [50, 90, 99, 100])))
ncu = len(set(cu.tolist()))
busy = {}
for g in range(hb.num_graphs):
    busy.setdefault(int(cu[g]), []).append((start[g], end[g]))
idle_tail = [end.max() - max(e for _, e in v) for v in busy.values()]
gaps = []
for v in busy.values():
    v.sort()
    for (s0, e0), (s1, e1) in zip(v, v[1:]):
        if s1 >= e0: gaps.append(s1 - e0)
print("CUs seen: %d; workgroups per CU: min %d max %d; idle behind a CU's last workgroup: mean %.1f us, max %.1f; gap between a CU's consecutive workgroups: %s" % (
    ncu, min(len(v) for v in busy.values()), max(len(v) for v in busy.values()), float(np.mean(idle_tail)), max(idle_tail),
    ("mean %.1f us, max %.1f (%d gaps)" % (float(np.mean(gaps)), max(gaps), len(gaps))) if gaps else "none (they overlap)"))
top = np.argsort(-end)[:8]
for g in top:
    print("  graph %4d: %3d vertices %6d entries  start %6.1f  duration %6.1f  end %6.1f" % (g, nv[g], ne[g], start[g], dur[g], end[g]))

names = ["P0a rowptr", "P0b entries", "P0c order", "first T", "first A", "hidden T (sum)", "barrier after T (sum)", "hidden A (sum)", "barrier after A (sum)", "last layer", "lgs", "tail"]
print("phase clocks of wave 0 by size class (mean microseconds; s_memtime / 2400):")
print("%-10s %6s " % ("vertices", "graphs") + " ".join("%9s" % n[:9] for n in names) + "   sum")
for n in sorted(set(nv.tolist())):
    idx = np.flatnonzero(nv == n)
    m = us[idx, :12].mean(axis=0)
    print("%-10d %6d " % (n, idx.size) + " ".join("%9.1f" % x for x in m) + "  %6.1f" % m.sum())
