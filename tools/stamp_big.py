#!/usr/bin/env python3
"""Phase clocks of k_big (needs the -DDGCN_DIAG build): DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_big.py er500|mc900 [layers=20] [graphs=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel
kind = sys.argv[1] if len(sys.argv) > 1 else "er500"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
if kind == "mc900":
    hb = datagen.multichannel_batch(B, 300, 0.03)
else:
    hb = datagen.er_batch(B, 500, 0.1)
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, 32), "cuda:0")
out = eng.solve_buffers(db, False)
for _ in range(20): eng.solve_fused(db, model, want_scores=False, out=out)
torch.cuda.synchronize()
st = torch.zeros(hb.num_graphs * 32, dtype=torch.int64, device="cuda")  # [graphs][16] phases, then [graphs][16] per-wave aggregation sums
_lib.set_option("diag_stamps", st.data_ptr())
eng.solve_fused(db, model, want_scores=False, out=out); torch.cuda.synchronize()
_lib.set_option("diag_stamps", 0)
allst = st.cpu().numpy().astype(np.float64)
raw = allst[:hb.num_graphs * 16].reshape(-1, 16)
waves = allst[hb.num_graphs * 16:].reshape(-1, 16)
names = ["row lengths, order, tiles, d^-1/2", "records", "layer 0 + transform of layer 1", "aggregations (sum)", "issuing the weight loads (sum)",
         "barrier behind the aggregation (sum)", "transforms (sum)", "barrier behind the transform (sum)", "last layer: walk, scores, priorities", "search, state, totals"]
wall = (raw[:, 13] - raw[:, 12]) / 100.0
ghz = raw[:, :12].sum(axis=1) / wall / 1e3
print("%s, %d graphs, %d layers: wave 0 of every workgroup, microseconds at the measured %.2f GHz: mean / max" % (kind, B, nl, ghz.mean()))
for i, nm in enumerate(names):
    us = raw[:, i] / (ghz.mean() * 1e3)
    print("%-42s %8.2f %8.2f" % (nm, us.mean(), us.max()))
print("%-42s %8.2f %8.2f" % ("workgroup wall time (100 MHz clock)", wall.mean(), wall.max()))
print("aggregation time of every wave (sum over the hidden layers, us, mean over graphs):")
print("  " + " ".join("%6.1f" % (waves[:, w].mean() / (ghz.mean() * 1e3)) for w in range(16)))
