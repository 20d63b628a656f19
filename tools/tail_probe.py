#!/usr/bin/env python3
"""Per-step cost of the tail kernel (csrc/tail.hip): 64 graphs of 64 vertices - the whole search runs in the tail - at
several model depths, against the step-by-step kernels; with the -DDGCN_DIAG build (DGCN_LIB=distgcn_amd/libdgcn_diag.so)
also with phases switched off (results meaningless then: the kernel's own step count divides its duration).
   python tools/tail_probe.py [graphs=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd import _lib
from distgcn_amd.api_common import get_engine
from distgcn_amd.mwis_gdpg_call import DQNAgent
from distgcn_amd.runtime_config import FLAGS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
diag = "diag" in os.environ.get("DGCN_LIB", "")
eng = get_engine()
hb = datagen.er_batch(B, 64, 0.05)
db = eng.upload(hb)
for layers in (3, 8, 20):
    agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=layers, diver_num=1, max_degree=1, predict="mwis"), seed=3)
    dm = agent.model.device_model(eng)
    for which, greedy in (("cit", eng.GREEDY_CENTRAL), ("rollout", eng.GREEDY_ROLLOUT)):
        line = {}
        for finish in (False, True):
            best, steps = 1e9, 0
            for rep in range(5):
                state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, finish_small=finish)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                best, steps = min(best, dt), res["steps"]
            line[finish] = (best, steps)
        s_full = line[False][1]
        print("layers %2d %-8s step by step: %3d steps %.3f ms (%.1f us per step) | with the tail: %d calls %.3f ms"
              % (layers, which, s_full, line[False][0] * 1e3, line[False][0] * 1e6 / max(s_full, 1), line[True][1], line[True][0] * 1e3), flush=True)
        for bits, name in ((0, "everything"), (1, "no aggregation"), (2, "no transform"), (4, "no weight fetch"), (7, "none of the three")) if diag else ((0, "everything"),):
            _lib.set_option("diag_flags", bits)
            out = eng.solve_buffers(db, False)
            best = None
            for rep in range(4):
                state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
                eng.timing(True)
                eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, finish_small=True, out=out, max_steps=2)
                torch.cuda.synchronize(); eng.timing(False)
                ms, n = eng.timing_read("tail_finish")
                steps = int(out["rounds"].cpu().numpy().max()) - 1  # (the call's own step counted one round)
                if n and steps > 0 and (best is None or ms / steps < best[0]):
                    best = (ms / steps, steps, ms)
            if best:
                print("      tail launch, %-20s %6.1f us per step of the longest graph (%d steps, %.3f ms)" % (name + ":", best[0] * 1e3, best[1], best[2]), flush=True)
        _lib.set_option("diag_flags", 0)
