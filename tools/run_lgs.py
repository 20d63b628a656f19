#!/usr/bin/env python3
"""Time the stand-alone local-greedy kernel: python tools/run_lgs.py [er|ba]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distgcn_amd import datagen
from distgcn_amd.engine import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "er"
hb = datagen.er_batch(500, 200, 0.1) if kind == "er" else datagen.ba_test2_batch(500)
eng = Engine("cuda:0"); db = eng.upload(hb)
for stats in (False, True):
    for _ in range(3): eng.lgs(db, prio=db.weights, want_stats=stats)
    torch.cuda.synchronize(); eng.timing(True)
    for _ in range(20): eng.lgs(db, prio=db.weights, want_stats=stats, sum_weights=db.weights)
    torch.cuda.synchronize(); eng.timing(False)
    ms, n = eng.timing_read("lgs")
    print("k_lgs %s stats=%s: %.1f us" % (kind, stats, ms / n * 1e3))
