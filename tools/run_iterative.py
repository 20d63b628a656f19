"""Time the iterative solvers (BASELINE config 5) on the device.

    python tools/run_iterative.py [--graphs 64] [--n 500] [--p 0.02] [--layers 20] [--beam 16]

Prints one JSON line per solver: steps (launches), wall time, graphs/s; `host` is the re-slicing
fallback path on the first graph only (what shapes outside the fused kernel get).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--p", type=float, default=0.02)
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--beam", type=int, default=16)
    ap.add_argument("--hidden", type=int, default=32, help="hidden width (narrower than 32: zero-padded onto the one-launch kernels beyond 512 vertices)")
    ap.add_argument("--host", type=int, default=1)
    ap.add_argument("--only", default=None, help="run this solver only (dit / cit / rollout)")
    ap.add_argument("--family", choices=["er", "mc"], default="er", help="mc: joint 3-channel conflict graphs of n // 3 flows (bench.multichannel_batch)")
    args = ap.parse_args()
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.api_common import get_engine
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from distgcn_amd.runtime_config import FLAGS
    eng = get_engine()
    if args.family == "mc":
        hb = datagen.multichannel_batch(args.graphs, args.n // 3, args.p)
    else:
        hb = datagen.er_batch(args.graphs, args.n, args.p)
    flags = FLAGS.copy(feature_size=1, hidden1=args.hidden, num_layer=args.layers, diver_num=1, max_degree=1, predict="mwis")
    agent = DQNAgent(flags, seed=3)
    dm = agent.model.device_model(eng)
    db = eng.upload(hb)
    path = eng.solve_path(db, dm)
    assert path, "outside the device solvers"
    print(json.dumps({"path": {1: "fused kernel", 2: "any-size path (general.hip + big.hip)"}[path]}), flush=True)
    greedy = {"dit": eng.GREEDY_ROUNDS, "cit": eng.GREEDY_CENTRAL, "rollout": eng.GREEDY_ROLLOUT}
    for which in ("dit", "cit", "rollout"):
        if args.only and which != args.only:
            continue
        dt = None
        for rep in range(4):  # (the first run loads the kernels; the best of the others counts)
            state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = eng.solve_residual(db, dm, state, greedy=greedy[which], max_rounds=1, beam=args.beam)
            torch.cuda.synchronize()
            if rep:
                dt = time.perf_counter() - t0 if dt is None else min(dt, time.perf_counter() - t0)
        eng.check_status(res["status"])
        st = res["state"].cpu().numpy()
        tot = float(np.sum(hb.weights[st == 1]))
        line = {"solver": which, "graphs": args.graphs, "n": args.n, "layers": args.layers, "steps": res["steps"],
                "seconds": round(dt, 4), "graphs_per_s": round(args.graphs / dt, 2),
                "ms_per_step": round(1e3 * dt / max(res["steps"], 1), 4), "mean_total": round(tot / args.graphs, 4)}
        # one more search with the kernel families timed (HIP events round every launch): where a search's GPU time goes
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
        eng.timing(True)
        eng.solve_residual(db, dm, state, greedy=greedy[which], max_rounds=1, beam=args.beam)
        torch.cuda.synchronize()
        eng.timing(False)
        fams = {}
        for fam in ("fused_residual", "tail_finish", "big_residual", "big_forward", "general_prepare", "general_greedy", "lgs", "wide_residual",
                    "supports", "transform", "spmm", "layer"):
            ms, n = eng.timing_read(fam)
            if n:
                fams[fam] = {"ms": round(ms, 3), "launches": n}
        line["gpu_ms_by_family"] = fams
        if args.host:
            agent.device_iterative = False
            n0 = int(hb.graph_ptr[1])
            adj, w = hb.scipy_graph(0), hb.weights[:n0]
            fn = {"dit": agent.solve_mwis_dit, "cit": agent.solve_mwis_cit,
                  "rollout": lambda a, b: agent.solve_mwis_rollout(a, b, b=args.beam)}[which]
            t0 = time.perf_counter()
            got = fn(adj, w)
            line["host_path_seconds_one_graph"] = round(time.perf_counter() - t0, 3)
            agent.device_iterative = True
            sel = set(int(v) for v in np.flatnonzero(st[:n0] == 1))
            line["host_equals_device"] = bool(sel == got[0])
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
