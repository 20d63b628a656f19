import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd.api_common import get_engine
from distgcn_amd.mwis_gdpg_call import DQNAgent
from distgcn_amd.runtime_config import FLAGS
eng = get_engine()
hb = datagen.er_batch(64, 500, 0.1)
db = eng.upload(hb)
agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis"), seed=3)
dm = agent.model.device_model(eng)
for which, greedy in (("cit", eng.GREEDY_CENTRAL), ("rollout", eng.GREEDY_ROLLOUT)):
    for finish in (False, True):
        for rep in range(3):
            state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
            out = eng.solve_buffers(db, False)
            eng.timing(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = eng.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, finish_small=finish, out=out)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            eng.timing(False)
        fams = {f: eng.timing_read(f) for f in ("general_prepare", "big_forward", "big_solve", "general_greedy", "lgs", "tail_finish", "fused_residual")}
        print(which, "finish", finish, "calls", res["steps"], "ms %.3f" % (dt * 1e3), {k: (round(v[0], 3), v[1]) for k, v in fams.items() if v[1]}, flush=True)
