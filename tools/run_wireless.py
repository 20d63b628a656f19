#!/usr/bin/env python3
"""Wall time of the batched wireless scheduling simulation.
python tools/run_wireless.py [instances] [nflows] [timeslots]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from distgcn_amd import datagen, wireless
from distgcn_amd.mwis_dqn_call import DQNAgent
from distgcn_amd.runtime_config import FLAGS
I = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(5)
agent = DQNAgent(1, flags=FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis"))
adjs, traffics = [], []
for i in range(I):
    indptr, indices = datagen.er_graph(F, 0.1, rng)
    adjs.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(F, F)))
    traffics.append(wireless.make_traffic(F, T, 0.05 + 0.05 * (i % 10) / 10, seed=i))
for algo in ("Greedy", "DGCN-LGS"):
    wireless.simulate(adjs[:4], traffics[:4], algo=algo, agent=agent)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = wireless.simulate(adjs, traffics, algo=algo, agent=agent)
    dt = time.perf_counter() - t0
    avgq = np.mean([wireless.summarize(r)["avg_queue_len"] for r in res])
    print("%-9s %d instances x %d flows x %d slots: %.3f s  (%.0f instance-slots/s, mean queue %.2f)"
          % (algo, I, F, T, dt, I * (T - 1) / dt, avgq))
