#!/usr/bin/env python3
"""Wall time of the multi-channel scheduling simulation on joint conflict graphs (wireless_dqn_test_mc.py:159-289): K channels x
nflows flows = one K * nflows-vertex graph per instance, all five schedulers, every instance in lockstep on the device.
python tools/run_wireless_mc.py [instances] [nflows] [channels] [timeslots] [num_layer=20]
(num_layer=1: the reference launcher's own recipe, bash/twc_major_wireless_mc_test.sh:3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from distgcn_amd import datagen, wireless
from distgcn_amd.mwis_dqn_call import DQNAgent as DqnAgent
from distgcn_amd.mwis_gdpg_call import DQNAgent as GdpgAgent
from distgcn_amd.runtime_config import FLAGS
I = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = int(sys.argv[4]) if len(sys.argv) > 4 else 50
NL = int(sys.argv[5]) if len(sys.argv) > 5 else 20
flags = FLAGS.copy(feature_size=1, hidden1=32, num_layer=NL, diver_num=1, max_degree=1, predict="mwis")
adjs, traffics = [], []
for i in range(I):
    rng = np.random.default_rng(900 + i)
    indptr, indices = datagen.er_graph(F, 0.03, rng)
    base = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(F, F))
    chans = wireless.multichannel_conflict_simulate(base, k=K, p=0.8, rng=np.random.RandomState(i))
    adjs.append(wireless.multichannel_conflict_graph(chans)[1])
    traffics.append(wireless.make_traffic(F, T, 0.05 + 0.05 * (i % 10) / 10, n_ch=K, seed=i))
print("%d instances, joint graphs of %d x %d = %d vertices, %.1f entries per vertex, %d slots, num_layer=%d" % (I, K, F, K * F, adjs[0].nnz / (K * F), T, NL))
for algo in ("Greedy", "DGCN-LGS", "DGCN-LGS-it", "CGCN-CGS", "DGCN-RS"):
    agent = DqnAgent(1, flags=flags) if algo in ("Greedy", "DGCN-LGS") else GdpgAgent(flags, seed=3)
    wireless.simulate(adjs[:2], [dict(arrival_pkts=t["arrival_pkts"][:3], link_rates=t["link_rates"][:3]) for t in traffics[:2]], algo=algo, agent=agent)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = wireless.simulate(adjs, traffics, algo=algo, agent=agent)
    dt = time.perf_counter() - t0
    avgq = np.mean([wireless.summarize(r)["avg_queue_len"] for r in res])
    sched = np.mean([r["scheduled"][1:].mean() for r in res])
    print("%-12s %.3f s  (%.0f instance-slots/s, %.1f links scheduled per slot, mean queue %.2f)" % (algo, dt, I * (T - 1) / dt, sched, avgq), flush=True)
