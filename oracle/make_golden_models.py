#!/usr/bin/env python3
"""Generate tests/golden/all_models.npz: EVERY shipped checkpoint of the reference (``model/result_*``: 44 with
max_degree = 1, "cheb1", and the two "cheb2" ones with supports [I, L, L.L])
as float32 weights, with oracle scores on two fixture graphs.  TEST INFRASTRUCTURE; run only in the build
container (needs /root/reference):

    python oracle/make_golden_models.py

Contents (data only): ``names``; ``<model>|<variable>`` float32 tensors read from the TF V2 bundles (A11);
``<model>|meta`` = [feature_size, hidden, num_layer, max_degree]; per graph g in GRAPHS: ``g%02d|<model>|f32`` / ``|f64``
scores of the oracle restatement (oracle/ref_numpy.py - NOT reference output: TensorFlow cannot run here) and
``|set`` / ``|rounds`` of the reference's own ``local_greedy_search_count`` on the f32 priorities.
For the ``cheb2`` checkpoints the state holds the imported-reference-equivalent ``simple_polynomials(adj, 2)``
(oracle/ref_numpy.py, itself pinned against the imported reference by tests/golden/supports.npz ``*_lap2_*``).
"""
import os
import re
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, import_reference  # noqa: E402
from distgcn_amd.checkpoint import load_bundle  # noqa: E402
from oracle import ref_numpy as orc  # noqa: E402

GRAPHS = [0, 9]  # fixture graph ids of tests/golden/graphs.npz (an ER and a BA graph)
NAME_RE = re.compile(r"^result_(?P<set>\w+?)_deep_ld(?P<ld>\d+)_c(?P<c>\d+)_l(?P<l>\d+)_cheb(?P<k>\d+)_diver(?P<d>\d+)_(?P<p>mwis|mis)_dqn$")


def main():
    ref_h, _ = import_reference()[:2]
    graphs = np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz"))
    out, names, skipped = {}, [], []
    for m in sorted(os.listdir(os.path.join(REF, "model"))):
        mm = NAME_RE.match(m)
        if not mm:
            continue
        md = int(mm.group("k"))
        if md not in (1, 2):
            skipped.append(m)
            continue
        tensors = load_bundle(os.path.join(REF, "model", m))
        params = {k: v for k, v in tensors.items() if "Adam" not in k and not k.endswith("_power")}
        layers = orc.gcn_layer_specs(params, num_supports=md + 1)
        assert "gcn_dqn/graphconvolution_1_vars/weights_%d" % (md + 1) not in params, m
        F = layers[0]["weights"][0].shape[0]
        # the directory name is not authoritative: result_IS4SAT_deep_ld1_c1_l1_cheb2_* holds [32, 1] weights
        assert F == int(mm.group("ld")) or md == 2, (m, F)
        assert len(layers) == int(mm.group("l")), (m, len(layers))
        names.append(m)
        for k, v in params.items():
            out["%s|%s" % (m, k)] = np.asarray(v, dtype=np.float32)
        out["%s|meta" % m] = np.array([F, int(mm.group("c")), len(layers), md], dtype=np.int32)
        predict = mm.group("p")
        for gi in GRAPHS:
            key = "g%02d" % gi
            w = graphs[key + "_weights"]
            n = w.size
            adj = sp.csr_matrix((np.ones(graphs[key + "_indices"].size), graphs[key + "_indices"],
                                 graphs[key + "_indptr"]), shape=(n, n))
            state = orc.makestate(adj, w.reshape(n, 1), F, md, "dqn_call")  # rows of 1/F (weights are > 0)
            s32, _ = orc.gcn_forward(layers, state, np.float32)
            s64, _ = orc.gcn_forward(layers, state, np.float64)
            out["%s|%s|f32" % (key, m)] = s32.ravel()
            out["%s|%s|f64" % (key, m)] = s64.ravel()
            prio = orc.priority(s32, w, predict)
            sref, _, rref = ref_h.local_greedy_search_count(adj, prio)
            out["%s|%s|set" % (key, m)] = np.array(sorted(sref), dtype=np.int32)
            out["%s|%s|rounds" % (key, m)] = np.int32(rref)
    out["names"] = np.array(names)
    out["skipped"] = np.array(skipped)
    out["graphs"] = np.array(GRAPHS, dtype=np.int32)
    path = os.path.join(ROOT, "tests", "golden", "all_models.npz")
    np.savez_compressed(path, **out)
    print("%d models (%d skipped: %s), %d bytes" % (len(names), len(skipped), ", ".join(skipped), os.path.getsize(path)))


if __name__ == "__main__":
    main()
