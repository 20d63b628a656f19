#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own functions.  TEST INFRASTRUCTURE.

Run only in the build container (it needs /root/reference, which never travels to the GPU box):

    python oracle/make_golden.py

What is produced (all small, data only - no reference source text):

* ``graphs.npz``      - fixture graphs from the reference's shipped ``data/*test2`` folders as
                         CSR + float64 weights + the ``greedy_utility`` / ``mwis_utility`` the
                         reference stored in each ``.mat`` (Data_Generation.py:218-219).
* ``supports.npz``    - output of the imported ``gcn.utils.simple_polynomials`` /
                         ``preprocess_features`` for every fixture graph (pins A1-A3).
* ``lgs.npz``         - outputs of the imported ``heuristics.local_greedy_search_{stats,overhead,
                         nstep}`` and ``greedy_search`` for several priority vectors per graph,
                         including heavy ties, negatives and zeros (pins A8, A8', A9).
* ``models.npz``      - float32 weights read from the shipped TF checkpoints (A11).
* ``scores.npz``      - GCN scores from the *oracle restatement* (oracle/ref_numpy.py) in f32 and
                         f64.  NOT reference output: TensorFlow cannot run here, so these are
                         restatement-derived ("parity unpinned") and labelled as such.

The reference's ``heuristics.py`` imports dwave_networkx / igraph / pulp at module level although
the greedy functions use none of them; empty stand-in modules are registered *for the import
only* (they are never called).
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DGCN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def import_reference():
    for name in ("dwave_networkx", "igraph", "pulp"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            if name == "pulp":
                mod.GLPK = object
            sys.modules[name] = mod
    if not hasattr(np, "bool"):
        np.bool = bool  # gcn/utils.py:25 touches np.bool in dead code on old NumPy only
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "gcn"))
    with contextlib.redirect_stdout(io.StringIO()):
        import heuristics as ref_h  # prints the networkx version on import
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gcn_utils", os.path.join(REF, "gcn", "utils.py"))
    ref_u = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_u)
    return ref_h, ref_u


FIXTURE_FILES = [
    ("ER_Graph_Uniform_GEN21_test2", "ER_n200_p0.1_b0_uni.mat"),     # config C1 / SURVEY anchors
    ("ER_Graph_Uniform_GEN21_test2", "ER_n100_p0.02_b0_uni.mat"),    # many isolated vertices
    ("ER_Graph_Uniform_GEN21_test2", "ER_n100_p0.1_b3_uni.mat"),
    ("ER_Graph_Uniform_GEN21_test2", "ER_n150_p0.1_b1_uni.mat"),
    ("ER_Graph_Uniform_GEN21_test2", "ER_n250_p0.04_b2_uni.mat"),
    ("ER_Graph_Uniform_GEN21_test2", "ER_n300_p0.05_b4_uni.mat"),
    ("ER_Graph_Uniform_GEN21_test2", "ER_n200_p0.1_b7_uni.mat"),
    ("BA_Graph_Uniform_GEN21_test2", None),  # filled in below from the directory listing
]

MODELS = [
    "result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn",
    "result_IS4SAT_deep_ld1_c32_l3_cheb1_diver1_mwis_dqn",
    "result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn",
    "result_DQNBA_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn",
    "result_DQNBA_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn",
    "result_DQNMED_deep_ld1_c16_l1_cheb1_diver1_mwis_dqn",   # carries a bias
    "result_IS4SAT_deep_ld1_c16_l4_cheb1_diver1_mwis_dqn",
]


def pick_files():
    files = [f for f in FIXTURE_FILES if f[1] is not None]
    for folder in ("ER_Graph_Uniform_GEN21_test2", "BA_Graph_Uniform_GEN21_test2"):
        names = sorted(os.listdir(os.path.join(REF, "data", folder)))
        if folder.startswith("BA"):
            want = ["n100_p0.02", "n100_p0.1", "n150_p0.1", "n200_p0.1", "n250_p0.04", "n300_p0.05", "n300_p0.0667"]
            for w in want:
                hit = [n for n in names if "_%s_" % w in n]
                if hit:
                    files.append((folder, hit[0]))
    return files


def priority_variants(w, rng):
    """Priority vectors that exercise every branch of the LGS rule."""
    n = w.size
    out = {"raw": w.copy()}
    out["ties1"] = np.round(w, 1)                       # heavy ties -> index tie-break
    out["ties0"] = np.ones(n)                            # everything equal
    neg = w - 0.5
    neg[rng.random(n) < 0.1] = 0.0                       # signed, with exact zeros (+0.0 / -0.0)
    neg[rng.random(n) < 0.05] = -0.0
    out["signed"] = neg
    out["int3"] = rng.integers(0, 3, size=n).astype(np.float64)
    return out


def main():
    ref_h, ref_u = import_reference()
    from oracle import ref_numpy as orc
    from distgcn_amd.checkpoint import load_bundle

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20230600)
    files = pick_files()

    graphs, supports, lgs = {}, {}, {}
    names = []
    for gi, (folder, fname) in enumerate(files):
        mat = sio.loadmat(os.path.join(REF, "data", folder, fname))
        adj = sp.csr_matrix(mat["adj"])
        adj.sort_indices()
        w = np.asarray(mat["weights"], dtype=np.float64).ravel()
        key = "g%02d" % gi
        names.append(fname)
        graphs[key + "_indptr"] = adj.indptr.astype(np.int32)
        graphs[key + "_indices"] = adj.indices.astype(np.int32)
        graphs[key + "_weights"] = w
        graphs[key + "_greedy_utility"] = np.float64(mat["greedy_utility"].ravel()[0])
        graphs[key + "_mwis_utility"] = np.float64(mat["mwis_utility"].ravel()[0])

        # ---- A1-A3 from the imported reference
        sup = ref_u.simple_polynomials(adj, 1)
        lap = sp.csr_matrix((sup[1][1], (sup[1][0][:, 0], sup[1][0][:, 1])), shape=sup[1][2])
        lap.sort_indices()
        supports[key + "_lap_indptr"] = lap.indptr.astype(np.int32)
        supports[key + "_lap_indices"] = lap.indices.astype(np.int32)
        supports[key + "_lap_data"] = lap.data.astype(np.float64)
        if w.size <= 100:  # L^2 is nearly dense: keep it for the small graphs only
            sup2 = ref_u.simple_polynomials(adj, 2)
            lap2 = sp.csr_matrix((sup2[2][1], (sup2[2][0][:, 0], sup2[2][0][:, 1])), shape=sup2[2][2])
            lap2.sort_indices()
            supports[key + "_lap2_indptr"] = lap2.indptr.astype(np.int32)
            supports[key + "_lap2_indices"] = lap2.indices.astype(np.int32)
            supports[key + "_lap2_data"] = lap2.data.astype(np.float64)
        feats = ref_u.preprocess_features(sp.lil_matrix(np.ones([w.size, 1]) * w[:, None]))
        dense = np.zeros(w.size)
        dense[feats[0][:, 0]] = feats[1]
        supports[key + "_feat_rownorm"] = dense

        # ---- A8 / A8' / A9 from the imported reference
        for vname, prio in priority_variants(w, rng).items():
            k2 = "%s_%s" % (key, vname)
            s, tot, rounds, p2p, bst, oh = ref_h.local_greedy_search_overhead(adj, prio)
            s2, tot2 = ref_h.local_greedy_search(adj, prio)
            s3, tot3, rounds3 = ref_h.local_greedy_search_count(adj, prio)
            s4, tot4, rounds4, p2p4, bst4 = ref_h.local_greedy_search_stats(adj, prio)
            assert s == s2 == s3 == s4 and rounds == rounds3 == rounds4 and (p2p, bst) == (p2p4, bst4)
            lgs[k2 + "_prio"] = prio
            lgs[k2 + "_set"] = np.array(sorted(s), dtype=np.int32)
            lgs[k2 + "_total"] = np.float64(tot)
            lgs[k2 + "_rounds"] = np.int32(rounds)
            lgs[k2 + "_p2p"] = np.int64(p2p)
            lgs[k2 + "_bst"] = np.int64(bst)
            lgs[k2 + "_overhead"] = np.asarray(oh, dtype=np.float64)
            for ns in (1, 2):
                sn, totn, nbn = ref_h.local_greedy_search_nstep(adj, prio, nstep=ns)
                lgs["%s_n%d_set" % (k2, ns)] = np.array(sorted(sn), dtype=np.int32)
                lgs["%s_n%d_nb" % (k2, ns)] = np.array(sorted(nbn), dtype=np.int32)
                lgs["%s_n%d_total" % (k2, ns)] = np.float64(totn)
            if vname == "raw":  # distinct weights: argsort order is well defined
                gs, gtot = ref_h.greedy_search(adj, prio)
                lgs[k2 + "_greedy_set"] = np.array(sorted(gs), dtype=np.int32)
                lgs[k2 + "_greedy_total"] = np.float64(gtot)
    graphs["names"] = np.array(names)

    # ---- A11 weights from the shipped checkpoints
    models = {}
    for m in MODELS:
        tensors = load_bundle(os.path.join(REF, "model", m))
        for name, arr in tensors.items():
            if "Adam" in name or name.endswith("_power"):
                continue
            models["%s|%s" % (m, name)] = arr
    models["names"] = np.array(MODELS)

    # ---- restatement-derived scores (NOT reference output; see module docstring)
    scores = {}
    for gi in range(len(files)):
        key = "g%02d" % gi
        n = graphs[key + "_weights"].size
        adj = sp.csr_matrix((np.ones(graphs[key + "_indices"].size), graphs[key + "_indices"],
                             graphs[key + "_indptr"]), shape=(n, n))
        w = graphs[key + "_weights"]
        for m in MODELS:
            params = {k.split("|", 1)[1]: v for k, v in models.items() if k.startswith(m + "|")}
            layers = orc.gcn_layer_specs(params)
            state = orc.makestate(adj, w.reshape(n, 1), 1, 1, "gdpg")
            s32, _ = orc.gcn_forward(layers, state, np.float32)
            s64, _ = orc.gcn_forward(layers, state, np.float64)
            scores["%s|%s|f32" % (key, m)] = s32.ravel()
            scores["%s|%s|f64" % (key, m)] = s64.ravel()
            prio = orc.priority(s32, w)
            st, rounds = orc.lgs_vectorised(adj.indptr, adj.indices, prio)
            sref, _, rref = ref_h.local_greedy_search_count(adj, prio)
            assert set(np.flatnonzero(st == 1)) == sref and rounds == rref
            scores["%s|%s|set" % (key, m)] = np.array(sorted(sref), dtype=np.int32)
            scores["%s|%s|rounds" % (key, m)] = np.int32(rref)

    np.savez_compressed(os.path.join(OUT, "graphs.npz"), **graphs)
    np.savez_compressed(os.path.join(OUT, "supports.npz"), **supports)
    np.savez_compressed(os.path.join(OUT, "lgs.npz"), **lgs)
    np.savez_compressed(os.path.join(OUT, "models.npz"), **models)
    np.savez_compressed(os.path.join(OUT, "scores.npz"), **scores)
    for f in sorted(os.listdir(OUT)):
        print("%-16s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
