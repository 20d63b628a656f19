#!/usr/bin/env python3
"""Time of ONE residual-graph step against the number of remaining vertices: a rollout / cit search run step by step with
an event pair around every launch (product build).   python tools/step_curve.py [rollout|cit] [graphs=64] [n=500]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.api_common import get_engine
from distgcn_amd.mwis_gdpg_call import DQNAgent
from distgcn_amd.runtime_config import FLAGS

which = sys.argv[1] if len(sys.argv) > 1 else "rollout"
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
eng = get_engine()
hb = datagen.er_batch(graphs, n, 0.02)
agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis"), seed=3)
dm = agent.model.device_model(eng)
db = eng.upload(hb)
greedy = eng.GREEDY_ROLLOUT if which == "rollout" else eng.GREEDY_CENTRAL
rows = []
for rep in range(2):
    state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
    rows = []
    for step in range(400):
        left = int((state == 0).sum().item())
        if left == 0:
            break
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, max_steps=1)
        e1.record()
        torch.cuda.synchronize()
        rows.append((step, left / graphs, e0.elapsed_time(e1) * 1e3))
print("%s search, %d graphs of %d vertices: step, vertices left per graph (mean), launch + kernel time of the step (us, event pair)" % (which, graphs, n))
for step, left, us in rows:
    if step % 5 == 0 or step == len(rows) - 1:
        print("%4d  %6.1f  %7.1f" % (step, left, us))
a = np.array(rows)
A = np.stack([np.ones(len(a)), a[:, 1]], 1)
c, *_ = np.linalg.lstsq(A, a[:, 2], rcond=None)
print("least squares: %.1f us + %.3f us per remaining vertex; total %.2f ms over %d steps" % (c[0], c[1], a[:, 2].sum() / 1e3, len(a)))
