#!/usr/bin/env python3
"""Phase clocks of ONE step of the residual-graph kernel in the middle of a rollout search (needs the -DDGCN_DIAG build):
   DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_residual.py [steps_before=40] [graphs=64] [n=500] [cit|rollout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen, _lib
from distgcn_amd.api_common import get_engine
from distgcn_amd.mwis_gdpg_call import DQNAgent
from distgcn_amd.runtime_config import FLAGS

before = int(sys.argv[1]) if len(sys.argv) > 1 else 40
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
which = sys.argv[4] if len(sys.argv) > 4 else "rollout"
eng = get_engine()
hb = datagen.er_batch(graphs, n, 0.02)
agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis"), seed=3)
if os.environ.get("STAMP_NOBIAS") == "1":  # the bench's C5 model (IS4SAT weights) has no biases
    for lyr in agent.model.layers:
        lyr["bias"] = None
    agent.model._device_model = None
dm = agent.model.device_model(eng)
db = eng.upload(hb)
greedy = eng.GREEDY_ROLLOUT if which == "rollout" else eng.GREEDY_CENTRAL
state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
if before > 0:
    eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, max_steps=before)
torch.cuda.synchronize()
left = (state.cpu().numpy() == 0).reshape(graphs, n).sum(1)
st = torch.zeros(graphs * 64, dtype=torch.int64, device="cuda")
_lib.set_option("diag_stamps", st.data_ptr())
eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, max_steps=1)
torch.cuda.synchronize()
_lib.set_option("diag_stamps", 0)
s = st.cpu().numpy().reshape(-1, 64).astype(np.float64) / 100.0
names = ["P0 states, anything left, renumbering", "P0 row bounds, columns, counts, slots", "P0 entries, row order, records", "first T", "first A", "hidden T (sum)", "barrier after T (sum)",
         "hidden A (sum)", "barrier after A (sum)", "last layer", "priorities, ranks, candidates", "rounds / completions, pick, output"]
print("%s step after %d steps: %d graphs of %d vertices, %.0f left on average (min %d, max %d)" % (which, before, graphs, n, left.mean(), left.min(), left.max()))
print("phase clocks of wave 0, hundred cycles: mean over graphs / max")
for i, nm in enumerate(names):
    print("%-40s %8.2f %8.2f" % (nm, s[:, i].mean(), s[:, i].max()))
print("%-40s %8.2f" % ("sum of means", s[:, :12].mean(axis=0).sum()))
raw = st.cpu().numpy().reshape(-1, 64)
wall_us = (raw[:, 15] - raw[:, 14]) / 100.0  # s_memrealtime: 100 MHz
ok = wall_us > 0
if ok.any():
    cyc = raw[ok, :12].sum(axis=1)
    print("workgroup wall time (100 MHz clock): mean %.1f us, max %.1f us; phase clocks tick at %.2f GHz" % (wall_us[ok].mean(), wall_us[ok].max(), (cyc / wall_us[ok]).mean() / 1e3))
blk = raw[:, 48:54].astype(np.float64)
if blk[:, 4].sum() > 0:
    nb, ntr = blk[:, 4].mean(), blk[:, 5].mean()
    print("wave 0's hidden aggregations, per graph: %.1f row blocks, %.1f trips; cycles per block: prologue %.0f, tails %.0f, epilogue %.0f; per trip %.0f"
          % (nb, ntr, blk[:, 0].mean() / nb, blk[:, 2].mean() / nb, blk[:, 3].mean() / nb, blk[:, 1].mean() / max(ntr, 1)))
