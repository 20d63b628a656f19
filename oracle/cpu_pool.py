"""TEST/BENCH INFRASTRUCTURE - the all-cores CPU figure of SURVEY 8d ("multiprocessing.Pool(nproc) over graphs
as the generous all-cores figure").  Runs the oracle restatement (oracle/ref_numpy.solve_mwis_gdpg, one graph per
call like the reference) in `procs` forked workers for `seconds` and prints one JSON line.

Started by bench.py's cpu_baseline leg as a CHILD PROCESS that never touches the GPU:
    python oracle/cpu_pool.py <graphs> <n> <p> <layers> <hidden> <seconds> <procs> <models.npz or ->
"""
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

_G = {}


def _work(args):
    wid, seconds = args
    from oracle import ref_numpy as orc
    hb, layers = _G["hb"], _G["layers"]
    slices = hb.graph_slices()
    done, g = 0, (wid * 7) % hb.num_graphs
    t0 = time.perf_counter()
    c0 = time.process_time()
    while time.perf_counter() - t0 < seconds:
        n0, n1 = slices[g]
        orc.solve_mwis_gdpg(layers, hb.scipy_graph(g), hb.weights[n0:n1], feature_size=1)
        done += 1
        g = (g + 1) % hb.num_graphs
    return done, time.perf_counter() - t0, time.process_time() - c0


def main():
    graphs, n, p, nl, hidden, seconds, procs, models = sys.argv[1:9]
    graphs, n, nl, hidden, procs = int(graphs), int(n), int(nl), int(hidden), int(procs)
    p, seconds = float(p), float(seconds)
    from distgcn_amd import datagen
    from distgcn_amd.gcn.models import layers_from_params
    layers = None
    if models != "-" and os.path.isfile(models):
        z = np.load(models)
        pre = "result_IS4SAT_deep_ld1_c%d_l%d_cheb1_diver1_mwis_dqn|" % (hidden, nl)
        params = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        if params:
            layers = layers_from_params(params)
    if layers is None:
        layers = datagen.random_model(nl, hidden)
    _G["hb"] = datagen.er_batch(graphs, n, p)
    _G["layers"] = layers
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_work, [(i, seconds) for i in range(procs)], chunksize=1)
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    # rate = sum of the workers' own rates (pool start-up is not CPU work on the path)
    rate = sum(r[0] / r[1] for r in res)
    busy = sum(r[2] / r[1] for r in res)  # CPU-seconds per second the workers really got (cgroup limits show here)
    print(json.dumps({"value": rate, "unit": "graphs/s", "cores": procs, "solves": done, "wall_s": wall,
                      "effective_cores": busy}))


if __name__ == "__main__":
    main()
