#!/usr/bin/env python3
"""Phase clocks of the one-layer kernel (csrc/shallow.hip; needs the -DDGCN_DIAG build: DGCN_LIB=.../libdgcn_diag.so).
python tools/stamp_shallow.py [er100|er200|ba] [graphs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
kind = sys.argv[1] if len(sys.argv) > 1 else "er100"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 500
hb = datagen.er_batch(B, int(kind[2:] or 200), 0.1) if kind.startswith("er") else datagen.ba_test2_batch(B)
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(1, 32), "cuda:0")
for _ in range(5): eng.solve(db, model, mode=MODE_FUSED)
torch.cuda.synchronize()
st = torch.zeros(hb.num_graphs * 8, dtype=torch.int64, device="cuda")
_lib.set_option("diag_stamps", st.data_ptr())
eng.solve(db, model, mode=MODE_FUSED); torch.cuda.synchronize()
_lib.set_option("diag_stamps", 0)
s = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ["row pointers, weights (round trip 1)", "columns + degree table issued -> ids in LDS", "barrier after the LDS image",
         "entry values, chain, priority (+ NaN vote, barrier)", "greedy rounds", "outputs, totals"]
print("%s, %d graphs: phase clocks of thread 0 in shader cycles, mean over graphs / max" % (kind, B))
for i, n in enumerate(names):
    print("%-58s %8.0f %8.0f" % (n, s[:, i].mean(), s[:, i].max()))
print("%-58s %8.0f" % ("sum of means", s[:, :6].mean(axis=0).sum()))
t0, t1 = s[:, 7], s[:, 6]  # s_memrealtime at start / end: 100 MHz
print("workgroup life (100 MHz clock): mean %.2f us, max %.2f us; first start -> last end %.2f us; starts spread over %.2f us"
      % ((t1 - t0).mean() / 100, (t1 - t0).max() / 100, (t1.max() - t0.min()) / 100, (t0.max() - t0.min()) / 100))
eng.timing(True)
for _ in range(200): eng.solve(db, model, mode=MODE_FUSED)
torch.cuda.synchronize(); eng.timing(False)
ms, n = eng.timing_read("fused_solve")
print("kernel time of this build: %.1f us avg over %d launches" % (ms / n * 1e3, n))
