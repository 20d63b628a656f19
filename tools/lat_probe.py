#!/usr/bin/env python3
"""Where one solve_mwis call's time goes: the helper call alone (pack + launch + wait + result copies) vs the API call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distgcn_amd import datagen
from distgcn_amd.mwis_gdpg_call import DQNAgent as GAgent
from distgcn_amd.mwis_dqn_call import _host_solver
from distgcn_amd.api_common import get_engine
from distgcn_amd.runtime_config import FLAGS
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hb = datagen.er_batch(8, 200, 0.1)
flags = FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis")
ga = GAgent(flags, seed=3)
graphs = [(hb.scipy_graph(g), hb.weights[n0:n1]) for g, (n0, n1) in enumerate(hb.graph_slices())]
for i in range(10): ga.solve_mwis(*graphs[i % 8])
hs = _host_solver(get_engine(), ga.model, "mwis")
args = [([a.indptr], [a.indices], [np.ascontiguousarray(w, dtype=np.float64).ravel()]) for a, w in graphs]
def t(name, fn):
    t0 = time.perf_counter()
    for i in range(iters): fn(i % 8)
    print("%-40s %7.1f us" % (name, (time.perf_counter() - t0) / iters * 1e6))
t("_pyptr.solve_lists only", lambda i: hs._solve_lists(hs._fn[0], hs._fn[1], hs.handle.value, *args[i]))
t("HostSolver.solve", lambda i: hs.solve(*args[i]))
t("DQNAgent.solve_mwis", lambda i: ga.solve_mwis(*graphs[i]))
eng = get_engine()
from distgcn_amd.engine import DeviceModel
import torch
db = eng.upload(hb.subset(0, 1))
dm = ga.model.device_model(eng)
out = eng.solve_buffers(db, True)
for _ in range(20): eng.solve_fused(db, dm, out=out)
torch.cuda.synchronize(); eng.timing(True)
for _ in range(200): eng.solve_fused(db, dm, out=out)
torch.cuda.synchronize(); eng.timing(False)
ms, n = eng.timing_read("fused_solve"); print("kernel, back to back                     %7.1f us" % (ms / n * 1e3))
