#!/usr/bin/env python3
"""Latency of the per-graph drop-in calls (one graph per call, as the reference's scripts use them).
python tools/run_single.py [iters]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distgcn_amd import datagen, heuristics
from distgcn_amd.mwis_gdpg_call import DQNAgent as GAgent
from distgcn_amd.mwis_dqn_call import DQNAgent as DAgent
from distgcn_amd.runtime_config import FLAGS
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hb = datagen.er_batch(8, 200, 0.1)
flags = FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis")
ga = GAgent(flags, seed=3)
da = DAgent(1, flags=flags)
graphs = [(hb.scipy_graph(g), hb.weights[n0:n1]) for g, (n0, n1) in enumerate(hb.graph_slices())]

def timeit(name, fn):
    for i in range(5): fn(*graphs[i % 8])
    t0 = time.perf_counter()
    for i in range(iters): fn(*graphs[i % 8])
    dt = (time.perf_counter() - t0) / iters
    print("%-44s %8.1f us per call" % (name, dt * 1e6))

timeit("mwis_gdpg_call.DQNAgent.solve_mwis", ga.solve_mwis)
timeit("mwis_dqn_call.DQNAgent.solve_mwis", da.solve_mwis)
timeit("heuristics.local_greedy_search", heuristics.local_greedy_search)
timeit("heuristics.greedy_search", heuristics.greedy_search)
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for i in range(iters): ga.solve_mwis(*graphs[i % 8])
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
