#!/usr/bin/env python3
"""Generate tests/golden/dataset100.npz: 100 graphs of the reference's shipped test2 folders (every 10th file of
ER_Graph_Uniform_GEN21_test2 and BA_Graph_Uniform_GEN21_test2) with the known answers the reference stored in
them.  TEST INFRASTRUCTURE; run only in the build container (needs /root/reference):

    python oracle/make_golden_dataset.py

Per graph i: ``indptr_i`` int32, ``indices_i`` int16, ``weights_i`` float64 (``adj`` / ``weights`` of the .mat,
Data_Generation.py:218-219); arrays over the graphs: ``names``, ``greedy_utility`` (the reference's own
``greedy_search`` total, Data_Generation.py:149-153), ``mwis_utility``; and, computed HERE by importing the
reference's ``heuristics.py``: ``lgs_total`` / ``lgs_rounds`` of ``local_greedy_search_count`` on the raw weights.
``ratio|<model>`` = total weight of the oracle restatement's GCN + the reference's ``local_greedy_search`` divided
by ``greedy_utility`` (the p column of mwis_dqn_test.py:321) for four shipped checkpoints - restatement-derived
(TensorFlow cannot run here), anchored by SURVEY section 9's folder averages.
"""
import os
import sys

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, import_reference  # noqa: E402
from distgcn_amd.checkpoint import load_bundle  # noqa: E402
from oracle import ref_numpy as orc  # noqa: E402

MODELS = ["result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn", "result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn",
          "result_DQNBA_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn", "result_DQNBA_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn"]


def main():
    ref_h, _ = import_reference()
    out, names, gu, mu, lt, lr = {}, [], [], [], [], []
    layers = {}
    for m in MODELS:
        params = {k: v for k, v in load_bundle(os.path.join(REF, "model", m)).items() if "Adam" not in k and not k.endswith("_power")}
        layers[m] = orc.gcn_layer_specs(params)
    ratios = {m: [] for m in MODELS}
    i = 0
    for folder in ("ER_Graph_Uniform_GEN21_test2", "BA_Graph_Uniform_GEN21_test2"):
        files = sorted(os.listdir(os.path.join(REF, "data", folder)))[::10]
        for f in files:
            mat = sio.loadmat(os.path.join(REF, "data", folder, f))
            adj = sp.csr_matrix(mat["adj"])
            adj.sum_duplicates()
            adj.sort_indices()
            w = np.asarray(mat["weights"], dtype=np.float64).ravel()
            out["indptr_%d" % i] = adj.indptr.astype(np.int32)
            out["indices_%d" % i] = adj.indices.astype(np.int16)
            out["weights_%d" % i] = w
            names.append(folder.split("_")[0] + "/" + f)
            gu.append(float(mat["greedy_utility"].ravel()[0]))
            mu.append(float(mat["mwis_utility"].ravel()[0]))
            sol, total, rounds = ref_h.local_greedy_search_count(adj, w)
            lt.append(float(total))
            lr.append(int(rounds))
            for m in MODELS:
                state = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "gdpg")
                s32, _ = orc.gcn_forward(layers[m], state, np.float32)
                prio = orc.priority(s32, w)
                sel, _ = ref_h.local_greedy_search(adj, prio)
                ratios[m].append(float(np.sum(w[sorted(sel)])) / gu[-1])
            i += 1
    out["names"] = np.array(names)
    out["greedy_utility"] = np.array(gu)
    out["mwis_utility"] = np.array(mu)
    out["lgs_total"] = np.array(lt)
    out["lgs_rounds"] = np.array(lr, dtype=np.int32)
    for m in MODELS:
        out["ratio|" + m] = np.array(ratios[m])
        print("%-58s mean ratio ER %.4f  BA %.4f" % (m, np.mean(ratios[m][:50]), np.mean(ratios[m][50:])))
    path = os.path.join(ROOT, "tests", "golden", "dataset100.npz")
    np.savez_compressed(path, **out)
    print("%d graphs, %d bytes" % (i, os.path.getsize(path)))


if __name__ == "__main__":
    main()
