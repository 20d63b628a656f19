#!/usr/bin/env python3
"""Phase clocks of the fused kernel (needs the -DDGCN_DIAG build: DGCN_LIB=.../libdgcn_diag.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
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
s = st.cpu().numpy().reshape(-1, 64).astype(np.float64) / 2400.0  # s_memtime ticks at the shader clock (2.4 GHz under load: tools/stamp_residual.py) -> us
names = ["P0a rowptr", "P0b entries", "P0c order", "first T", "first A", "hidden T (sum)", "barrier after T (sum)",
         "hidden A (sum)", "barrier after A (sum)", "last layer", "lgs", "tail"]
print("phase clocks of wave 0, microseconds: mean over graphs / max")
cols = list(range(12))
for i, n in zip(cols, names):
    print("%-24s %8.2f %8.2f" % (n, s[:, i].mean(), s[:, i].max()))
print("%-24s %8.2f" % ("sum of means", s[:, cols].mean(axis=0).sum()))
blk = st.cpu().numpy().reshape(-1, 64)[:, 48:54].astype(np.float64)
if blk[:, 4].sum() > 0:
    nb, ntr = blk[:, 4].mean(), blk[:, 5].mean()
    print("last wave, per graph: %.1f row blocks, %.1f full trips; hundred cycles: prologue %.1f, trips %.1f, tails %.1f, epilogue %.1f"
          % (nb, ntr, blk[:, 0].mean() / 100, blk[:, 1].mean() / 100, blk[:, 2].mean() / 100, blk[:, 3].mean() / 100))
    print("  per block: prologue %.0f cycles, tails %.0f, epilogue %.0f; per full trip %.0f cycles"
          % (blk[:, 0].mean() / nb, blk[:, 2].mean() / nb, blk[:, 3].mean() / nb, blk[:, 1].mean() / ntr))
tot = s[:, cols].sum(axis=1)
print("per-workgroup total: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f;  first-dispatched half mean %.1f, second half mean %.1f"
      % (tot.min(), np.percentile(tot, 10), np.median(tot), np.percentile(tot, 90), tot.max(), tot[:256].mean(), tot[256:].mean()))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50): eng.solve(db, model, mode=MODE_FUSED)
ev1.record(); torch.cuda.synchronize()
print("launch time of this build: %.1f us" % (ev0.elapsed_time(ev1) * 1e3 / 50))

raw = st.cpu().numpy().reshape(-1, 64)
print("timeline of co-resident workgroups (block b and b + 256 share a CU), microseconds from the first stamp;")
print("per hidden layer: transform start-end | gather start-end")
for b in (0, 27, 31, 61):
    if b >= hb.num_graphs:
        continue
    t0 = raw[b, 16] if b + 256 >= hb.num_graphs else min(raw[b, 16], raw[b + 256, 16])
    for blk in ((b,) if b + 256 >= hb.num_graphs else (b, b + 256)):
        cells = []
        for l in range(6):
            ts = [(raw[blk, 16 + 4 * l + k] - t0) / 100.0 for k in range(4)]
            cells.append("T %5.1f-%5.1f A %5.1f-%5.1f" % tuple(ts))
        print("  block %3d: %s" % (blk, " | ".join(cells)))
