#!/usr/bin/env python3
"""tests/golden/multichannel.npz: the reference's OWN multi-channel conflict-graph functions, executed.  TEST INFRASTRUCTURE.

    python oracle/make_golden_multichannel.py        # needs /root/reference; never runs on the GPU box

``wireless_rollout_test_flood.py`` cannot be imported (it parses flags, loads TensorFlow models and imports modules the
reference does not ship), so the two function definitions are cut out of its syntax tree with ``ast`` at generation
time - ``poisson_multigraphs_from_dict`` (:70-95) and ``multichannel_conflict_graph`` (:98-133) - compiled as they
stand and run with NumPy / NetworkX; no reference text is stored.  Old-library spellings are aliased for the call only
(``nx.from_numpy_matrix``, ``nx.adjacency_matrix`` returning a SciPy matrix).  Stored: inputs (conflict graphs as CSR,
seeds, k, p) and outputs (per-channel adjacencies, the joint graph) as CSR arrays.
"""
import ast
import os
import sys

import networkx as nx
import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DGCN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "multichannel.npz")
CASES = [(0, 3, 0.8, 101), (2, 2, 0.5, 102), (12, 4, 0.9, 103), (1, 1, 1.0, 104)]  # (fixture graph, channels, p_overlap, seed)


def reference_functions():
    src = open(os.path.join(REF, "wireless_rollout_test_flood.py")).read()
    tree = ast.parse(src)
    wanted = {"poisson_multigraphs_from_dict", "multichannel_conflict_graph"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    if len(mod.body) != 2:
        raise RuntimeError("functions not found in the reference")
    if not hasattr(nx, "from_numpy_matrix"):
        nx.from_numpy_matrix = nx.from_numpy_array
    _adj = nx.adjacency_matrix
    shim_nx = type("NX", (), {})()
    for name in dir(nx):
        if not name.startswith("__"):
            setattr(shim_nx, name, getattr(nx, name))
    shim_nx.adjacency_matrix = lambda g, *a, **k: sp.csr_matrix(_adj(g, *a, **k))
    ns = {"np": np, "nx": shim_nx}
    exec(compile(mod, "<reference functions>", "exec"), ns)
    return ns["poisson_multigraphs_from_dict"], ns["multichannel_conflict_graph"]


def main():
    simulate, joint = reference_functions()
    z = np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz"))
    out = {"cases": np.array(CASES, dtype=np.float64)}
    for ci, (gi, k, p, seed) in enumerate(CASES):
        key = "g%02d" % gi
        n = z[key + "_weights"].size
        adj = sp.csr_matrix((np.ones(z[key + "_indices"].size), z[key + "_indices"], z[key + "_indptr"]), shape=(n, n))
        dense = np.asarray(adj.todense())
        gdict = {"adj_c": dense.copy(), "adj_i": dense.copy(), "xys": np.zeros((n, 2))}
        np.random.seed(seed)
        _, graphs = simulate(gdict, k=k, p=p)
        adj_list, adj_gk = joint(graphs)
        for c, a in enumerate(adj_list):
            a = sp.csr_matrix(a)
            a.sort_indices()
            out["c%d|ch%d|indptr" % (ci, c)] = a.indptr.astype(np.int64)
            out["c%d|ch%d|indices" % (ci, c)] = a.indices.astype(np.int64)
            out["c%d|ch%d|data" % (ci, c)] = a.data.astype(np.float64)
        a = sp.csr_matrix(adj_gk)
        a.sort_indices()
        out["c%d|joint|indptr" % ci] = a.indptr.astype(np.int64)
        out["c%d|joint|indices" % ci] = a.indices.astype(np.int64)
        out["c%d|joint|data" % ci] = a.data.astype(np.float64)
        print("case %d: graph g%02d n=%d k=%d p=%.2f -> joint %d vertices, %d entries" % (ci, gi, n, k, p, a.shape[0], a.nnz))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
