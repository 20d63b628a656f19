#!/usr/bin/env python3
"""Phase clocks of the fused kernel (needs the -DDGCN_DIAG build: DGCN_LIB=.../libdgcn_diag.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hb = datagen.er_batch(500, 200, 0.1) if kind == "er" else datagen.ba_test2_batch(500)
eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(nl, 32), "cuda:0")
for _ in range(3): eng.solve(db, model, mode=MODE_FUSED)
torch.cuda.synchronize()
st = torch.zeros(hb.num_graphs * 16, dtype=torch.int64, device="cuda")
os.environ["DGCN_FUSED_STAMPS"] = str(st.data_ptr())
eng.solve(db, model, mode=MODE_FUSED); torch.cuda.synchronize()
os.environ.pop("DGCN_FUSED_STAMPS")
s = st.cpu().numpy().reshape(-1, 16).astype(np.float64) / 100.0  # s_memtime ticks at 100 MHz -> us
names = ["P0a rowptr", "P0b entries", "P0c order", "first T", "first A", "hidden T (sum)", "barrier after T (sum)",
         "hidden A (sum)", "barrier after A (sum)", "last layer", "lgs", "tail"]
print("phase clocks of wave 0, microseconds: mean over graphs / max")
for i, n in enumerate(names):
    print("%-24s %8.2f %8.2f" % (n, s[:, i].mean(), s[:, i].max()))
print("%-24s %8.2f" % ("sum of means", s[:, :12].mean(axis=0).sum()))

raw = st.cpu().numpy().reshape(-1, 16)
hw = raw[:, 12]; xcc = raw[:, 13] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
ident = (xcc << 16) | (se << 8) | (sh << 4) | cu
import collections
groups = collections.defaultdict(list)
for b, i in enumerate(ident): groups[int(i)].append(b)
print("distinct (xcc,se,sh,cu):", len(groups), " blocks per CU histogram:", collections.Counter(len(v) for v in groups.values()))
print("first CUs -> blocks:", [groups[k] for k in sorted(groups)[:10]])
print("xcc of blocks 0..15:", xcc[:16].tolist())
