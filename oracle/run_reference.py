#!/usr/bin/env python3
"""Execute the REFERENCE'S OWN Python (mwis_dqn_call.py, mwis_gdpg_call.py, gcn/models.py, gcn/layers.py ...) in the
build container and store what it returns as golden vectors: tests/golden/ref_exec.npz.  TEST INFRASTRUCTURE.

    python oracle/run_reference.py            # needs /root/reference; never runs on the GPU box
    python oracle/run_reference.py --big      # tests/golden/ref_exec_big.npz: BA N=300 m=2 at l=20, one N=500 b=16 rollout

The reference imports TensorFlow, which cannot be installed here.  oracle/tf_shim/ puts a NumPy stand-in for the
~60 tf.compat.v1 entry points its model code touches first on sys.path (lazy graph nodes evaluated in float32 by
NumPy / SciPy; see that package's docstring), so the reference's files run UNMODIFIED from where they lie - nothing
is copied.  Old-library spellings the reference relies on are aliased for the import only (networkx's
from_scipy_sparse_matrix / adjacency_matrix -> SciPy matrices, np.bool) and three unused heavy imports are empty
stand-ins (dwave_networkx, igraph, pulp), exactly as oracle/make_golden.py does.

What these vectors pin (reference code executed, not restated): model assembly (layer count, widths, which layers
get which activation, bias, support slicing), checkpoint restore by variable name, makestate of both agents,
predict, the whole of DQNAgent.solve_mwis (zero-weight pruning with NetworkX, index mapping, local greedy search,
totals) and MWISSolver.solve_mwis / _dit / _cit / _rollout / _wrap control flow.  What they do NOT pin: the
arithmetic inside TensorFlow's kernels - the stand-in evaluates sparse_tensor_dense_matmul / matmul / add_n with
SciPy / NumPy float32, so the GCN scores here equal oracle/ref_numpy.py's float32 restatement whenever the two
agree on the STRUCTURE, and say nothing about TF's summation order ("parity unpinned" at that boundary remains).

Each configuration runs in a fresh interpreter (the reference builds its model and a module-level agent at import).
"""
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DGCN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "ref_exec.npz")

DQN_MODELS = [  # (checkpoint dir name, flags)
    ("result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn", dict(feature_size=1, hidden1=32, num_layer=20, max_degree=1)),
    ("result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn", dict(feature_size=1, hidden1=32, num_layer=1, max_degree=1)),
    ("result_DQNBA_deep_ld1_c32_l3_cheb1_diver1_mwis_dqn", dict(feature_size=1, hidden1=32, num_layer=3, max_degree=1)),
    ("result_IS4SAT_deep_ld1_c16_l4_cheb1_diver1_mwis_dqn", dict(feature_size=1, hidden1=16, num_layer=4, max_degree=1)),
    ("result_IS4SAT_deep_ld32_c32_l20_cheb1_diver1_mwis_dqn", dict(feature_size=32, hidden1=32, num_layer=20, max_degree=1)),
    ("result_IS4SAT_deep_ld32_c64_l2_cheb1_diver1_mwis_dqn", dict(feature_size=32, hidden1=64, num_layer=2, max_degree=1)),
    ("result_IS4SAT_deep_ld1_c1_l2_cheb2_diver1_mwis_dqn", dict(feature_size=1, hidden1=1, num_layer=2, max_degree=2)),
    ("result_DQNEPI_deep_ld1_c32_l20_cheb1_diver1_mis_dqn", dict(feature_size=1, hidden1=32, num_layer=20, max_degree=1, predict="mis")),
]
GRAPHS = [0, 2, 9, 12]  # fixture graph ids of tests/golden/graphs.npz
GDPG_CONFIGS = [dict(feature_size=1, hidden1=32, num_layer=3, max_degree=1, predict="mwis"),
                dict(feature_size=1, hidden1=32, num_layer=3, max_degree=1, predict="mis")]
GDPG_GRAPHS = [1, 7, 2]  # small graphs: the iterative solvers run hundreds of forward passes


def prepare_reference_imports():
    """sys.path + stand-ins so that the reference's modules import here."""
    for name in ("dwave_networkx", "igraph", "pulp"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            if name == "pulp":
                mod.GLPK = object
            sys.modules[name] = mod
    if not hasattr(np, "bool"):
        np.bool = bool
    if not hasattr(np, "float"):
        np.float = float
    import networkx as nx
    if not hasattr(nx, "from_scipy_sparse_matrix"):
        nx.from_scipy_sparse_matrix = nx.from_scipy_sparse_array
    _adj = nx.adjacency_matrix
    nx.adjacency_matrix = lambda g, *a, **k: sp.csr_matrix(_adj(g, *a, **k))  # networkx 2.x returned a SciPy *matrix*
    sys.path.insert(0, ROOT)  # distgcn_amd.checkpoint: the bundle reader behind the shim's Saver.restore
    sys.path.insert(0, os.path.join(REF, "gcn"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, "tf_shim"))


def fixture(i):
    z = np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz"))
    k = "g%02d" % i
    w = z[k + "_weights"]
    adj = sp.csr_matrix((np.ones(z[k + "_indices"].size), z[k + "_indices"], z[k + "_indptr"]), shape=(w.size, w.size))
    return adj, w


def flag_args(flags):
    base = dict(diver_num=1, epsilon=0.0002, wts_init="random")
    base.update(flags)
    return ["--%s=%s" % (k, v) for k, v in base.items()]


# ---------------------------------------------------------------------------------------------- workers
def worker_dqn(model_dir, out_path):
    prepare_reference_imports()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        import mwis_dqn_call as m  # builds placeholders, the model and `dqn_agent` at import
        m.dqn_agent.load(os.path.join(REF, "model", model_dir))
    import tensorflow as tf
    out = {"variable_names": np.array(sorted(tf.shim_variables()))}
    rng = np.random.default_rng(11)
    for gi in GRAPHS:
        adj, w = fixture(gi)
        state = m.dqn_agent.makestate(adj, w.reshape(-1, 1))
        act_values, action = m.dqn_agent.predict(state)
        out["g%02d|scores" % gi] = np.asarray(act_values, dtype=np.float32)
        out["g%02d|action" % gi] = np.asarray(action, dtype=np.int64)
        wz = w.copy()
        wz[rng.random(w.size) < 0.1] = 0.0  # the pruning branch of solve_mwis (mwis_dqn_call.py:202-207)
        for tag, ww in (("full", w), ("zeros", wz)):
            sol, total, reward = m.dqn_agent.solve_mwis(adj, ww, train=False)
            out["g%02d|%s|weights" % (gi, tag)] = ww
            out["g%02d|%s|set" % (gi, tag)] = np.array(sorted(int(v) for v in sol), dtype=np.int64)
            out["g%02d|%s|total" % (gi, tag)] = np.float64(total)
            out["g%02d|%s|reward" % (gi, tag)] = np.float64(reward)
    np.savez_compressed(out_path, **out)


def worker_gdpg(out_path):
    prepare_reference_imports()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        import mwis_gdpg_call as m
        agent = m.DQNAgent(m.FLAGS, 5000)
    import tensorflow as tf
    out = {}
    rng = np.random.RandomState(5)
    for name, val in tf.shim_variables().items():
        if name.startswith("model/") and name.endswith("/bias"):  # biases start at 0: make them matter
            for v in tf._variables:
                if v.name == name + ":0":
                    v.value_ = rng.uniform(-0.2, 0.2, size=np.shape(val)).astype(np.float32)
    for name, val in tf.shim_variables().items():
        if name.startswith("model/"):
            out["var|" + name] = np.asarray(val)
    for gi in GDPG_GRAPHS:
        adj, w = fixture(gi)
        state = agent.makestate(adj, w.reshape(-1, 1))
        act_values, action = agent.predict(state)
        out["g%02d|scores" % gi] = np.asarray(act_values, dtype=np.float32)
        out["g%02d|action" % gi] = np.asarray(action, dtype=np.int64)
        for which in ("solve_mwis", "solve_mwis_dit", "solve_mwis_cit", "solve_mwis_cgs_train", "solve_mwis_cit_wrap", "solve_mwis_rollout",
                      "solve_mwis_rollout_wrap", "solve_mwis_rollout00", "solve_mwis_rollout0", "solve_mwis_rollout1"):
            fn = getattr(agent, which, None)
            if fn is None:
                continue
            np.random.seed(1234)  # the rollouts break ties with np.random.choice
            kw = dict(b=8) if "rollout" in which else {}
            try:
                sol, total = fn(adj, w, train=False, **kw)
            except TypeError:
                sol, total = fn(adj, w, **kw)
            out["g%02d|%s|set" % (gi, which)] = np.array(sorted(int(v) for v in sol), dtype=np.int64)
            out["g%02d|%s|total" % (gi, which)] = np.float64(np.asarray(total).ravel()[0])
    np.savez_compressed(out_path, **out)


# ---- full-size cases (tests/golden/ref_exec_big.npz): where the float32 error tail lives and the C5 shape ----------
BIG_OUT = os.path.join(ROOT, "tests", "golden", "ref_exec_big.npz")
BIG_BA = [1320, 3945]       # graphs of datagen.ba_test2_batch (C4): N = 300, m = 2 - the two the round-2 review flagged
BIG_BA_MODEL = ("result_DQNBA_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn", dict(feature_size=1, hidden1=32, num_layer=20, max_degree=1))
BIG_C5 = dict(feature_size=1, hidden1=32, num_layer=20, max_degree=1, predict="mwis")  # + IS4SAT l=20 weights, ER G(500, 0.02), b = 16


def datagen_graph(kind, index):
    from distgcn_amd import datagen
    hb = datagen.ba_test2_batch(1, first_index=index) if kind == "ba" else datagen.er_batch(1, 500, 0.02, first_index=index)
    n = hb.num_nodes
    adj = sp.csr_matrix((np.ones(hb.col_idx.size), hb.col_idx, hb.row_ptr), shape=(n, n))
    return adj, hb.weights.copy()


def worker_big_ba(model_dir, out_path):
    """mwis_dqn_call.DQNAgent (DQNBA l=20) on BA N=300 m=2 graphs of the C4 batch: predict + solve_mwis."""
    prepare_reference_imports()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        import mwis_dqn_call as m
        m.dqn_agent.load(os.path.join(REF, "model", model_dir))
    out = {}
    for gi in BIG_BA:
        adj, w = datagen_graph("ba", gi)
        state = m.dqn_agent.makestate(adj, w.reshape(-1, 1))
        act_values, action = m.dqn_agent.predict(state)
        sol, total, _ = m.dqn_agent.solve_mwis(adj, w, train=False)
        out["ba%04d|scores" % gi] = np.asarray(act_values, dtype=np.float32)
        out["ba%04d|action" % gi] = np.asarray(action, dtype=np.int64)
        out["ba%04d|set" % gi] = np.array(sorted(int(v) for v in sol), dtype=np.int64)
        out["ba%04d|total" % gi] = np.float64(total)
    np.savez_compressed(out_path, **out)


def worker_big_c5(out_path):
    """mwis_gdpg_call.DQNAgent with 20 layers (the shipped IS4SAT l=20 weights copied into its variables, biases
    left at their initial 0) on one ER G(500, 0.02) graph: predict, solve_mwis_rollout(b=16) (mwis_gdpg_call.py:596-659),
    solve_mwis_cit and solve_mwis_cgs_train(train=False) (:778-839)."""
    prepare_reference_imports()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        import mwis_gdpg_call as m
        agent = m.DQNAgent(m.FLAGS, 5000)
    import tensorflow as tf
    z = np.load(os.path.join(ROOT, "tests", "golden", "all_models.npz"))
    pre = "result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn|gcn_dqn/"
    for v in tf._variables:
        name = v.name[:-2] if v.name.endswith(":0") else v.name
        for scope in ("model/gcn2_dqn/", "target/gcn2_dqn/"):
            if name.startswith(scope) and pre + name[len(scope):] in z.files:
                v.value_ = np.asarray(z[pre + name[len(scope):]], dtype=np.float32)
    out = {}
    for name, val in tf.shim_variables().items():
        if name.startswith("model/"):
            out["var|" + name] = np.asarray(val)
    adj, w = datagen_graph("er", 0)
    state = agent.makestate(adj, w.reshape(-1, 1))
    act_values, action = agent.predict(state)
    out["scores"] = np.asarray(act_values, dtype=np.float32)
    out["action"] = np.asarray(action, dtype=np.int64)
    for which, kw in (("solve_mwis", {}), ("solve_mwis_cit", {}), ("solve_mwis_cgs_train", {}), ("solve_mwis_rollout", dict(b=16))):
        np.random.seed(1234)
        sol, total = getattr(agent, which)(adj, w, train=False, **kw)
        out["%s|set" % which] = np.array(sorted(int(v) for v in sol), dtype=np.int64)
        out["%s|total" % which] = np.float64(np.asarray(total).ravel()[0])
    np.savez_compressed(out_path, **out)


def main_big():
    out = {"ba_graphs": np.array(BIG_BA), "ba_model": np.array(BIG_BA_MODEL[0]), "c5_flags": np.array(json.dumps(BIG_C5))}
    res = run_worker("big_ba", [BIG_BA_MODEL[0]], BIG_BA_MODEL[1])
    out.update({"ba|" + k: v for k, v in res.items()})
    print("big_ba ok (%d arrays)" % len(res))
    res = run_worker("big_c5", [], BIG_C5)
    out.update({"c5|" + k: v for k, v in res.items()})
    print("big_c5 ok (%d arrays): rollout set of %d vertices, total %.6f" % (len(res), res["solve_mwis_rollout|set"].size, float(res["solve_mwis_rollout|total"])))
    np.savez_compressed(BIG_OUT, **out)
    print("%s: %d arrays, %d bytes" % (BIG_OUT, len(out), os.path.getsize(BIG_OUT)))


def worker_test_loop(out_path):
    """The reference's evaluation script itself, mwis_dqn_test.py:304-348 (A12): run with runpy from a scratch
    directory holding ./model -> the reference's model/ and ./data/<folder> -> symlinks to the 50 .mat files of
    tests/golden/dataset100.npz, so that the reference tree is never written to.  Collects the ./output/*.csv it writes."""
    prepare_reference_imports()
    import contextlib, io, runpy
    import pandas as pd
    if not hasattr(pd.DataFrame, "append"):  # pandas < 2 spelling used at mwis_dqn_test.py:340
        pd.DataFrame.append = lambda self, other, ignore_index=False: pd.concat([self, pd.DataFrame([other])], ignore_index=ignore_index)
    import scipy.io as sio
    _loadmat = sio.loadmat

    def loadmat_csc(*a, **k):  # SciPy of the reference's time handed sparse variables out as csc_matrix, not COO
        m = _loadmat(*a, **k)
        return {key: (sp.csc_matrix(v) if sp.issparse(v) else v) for key, v in m.items()}
    sio.loadmat = loadmat_csc
    np.random.seed(20230601)  # the script draws np.random.permutation and, per graph, one np.random.rand() (epsilon branch)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(REF, "mwis_dqn_test.py"), run_name="__main__")
    csvs = [f for f in os.listdir("output") if f.endswith(".csv")]
    df = pd.read_csv(os.path.join("output", csvs[0]))
    fired = sum(1 for ln in buf.getvalue().splitlines() if "Ratio" in ln)
    np.savez_compressed(out_path, data=np.array([str(x) for x in df["data"]]), p=df["p"].to_numpy(dtype=np.float64),
                        printed=np.int64(fired))


TEST_LOOP = [("IS4SAT", 1), ("IS4SAT", 20), ("DQNBA", 1), ("DQNBA", 20)]  # bash/generalization_dqn_test.sh:20-33


def run_test_loop(training_set, num_layer, family):
    """-> {file name: p} of the reference's own mwis_dqn_test.py on dataset100's graphs of one family."""
    names = [str(n) for n in np.load(os.path.join(ROOT, "tests", "golden", "dataset100.npz"))["names"] if str(n).startswith(family + "/")]
    folder = "%s_Graph_Uniform_GEN21_test2" % family
    with tempfile.TemporaryDirectory() as d:
        os.symlink(os.path.join(REF, "model"), os.path.join(d, "model"))
        os.makedirs(os.path.join(d, "data", folder))
        os.makedirs(os.path.join(d, "output"))
        for n in names:
            os.symlink(os.path.join(REF, "data", folder, n.split("/", 1)[1]), os.path.join(d, "data", folder, n.split("/", 1)[1]))
        path = os.path.join(d, "o.npz")
        flags = ["--training_set=%s" % training_set, "--epsilon=.0002", "--feature_size=1", "--diver_num=1",
                 "--datapath=./data/%s" % folder, "--max_degree=1", "--predict=mwis", "--learning_rate=0.00001", "--hidden1=32",
                 "--num_layer=%d" % num_layer, "--epochs=10"]
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "test_loop", path] + flags
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError("reference mwis_dqn_test.py failed:\n%s" % r.stderr[-4000:])
        z = np.load(path)
        return {family + "/" + str(k): float(v) for k, v in zip(z["data"], z["p"])}


# ---------------------------------------------------------------------------------------------- driver
def run_worker(kind, extra, flags):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "o.npz")
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", kind, path] + extra + flag_args(flags)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=REF)
        if r.returncode != 0:
            raise RuntimeError("reference worker %s failed:\n%s" % (kind, r.stderr[-4000:]))
        z = np.load(path)
        return {k: z[k] for k in z.files}


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--worker":
        kind, path = sys.argv[2], sys.argv[3]
        rest = sys.argv[4:]
        if kind == "dqn":
            model_dir = rest[0]
            sys.argv = [sys.argv[0]] + rest[1:]
            worker_dqn(model_dir, path)
        elif kind == "test_loop":
            sys.argv = [sys.argv[0]] + rest
            worker_test_loop(path)
        elif kind == "big_ba":
            model_dir = rest[0]
            sys.argv = [sys.argv[0]] + rest[1:]
            worker_big_ba(model_dir, path)
        elif kind == "big_c5":
            sys.argv = [sys.argv[0]] + rest
            worker_big_c5(path)
        else:
            sys.argv = [sys.argv[0]] + rest
            worker_gdpg(path)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--big":
        return main_big()
    out = {"dqn_models": np.array([m for m, _ in DQN_MODELS]), "graphs": np.array(GRAPHS), "gdpg_graphs": np.array(GDPG_GRAPHS),
           "dqn_flags": np.array([json.dumps(f) for _, f in DQN_MODELS]), "gdpg_flags": np.array([json.dumps(f) for f in GDPG_CONFIGS])}
    for model_dir, flags in DQN_MODELS:
        res = run_worker("dqn", [model_dir], flags)
        for k, v in res.items():
            out["dqn|%s|%s" % (model_dir, k)] = v
        print("dqn  %-60s ok (%d arrays)" % (model_dir, len(res)))
    for ci, flags in enumerate(GDPG_CONFIGS):
        res = run_worker("gdpg", [], flags)
        for k, v in res.items():
            out["gdpg|%d|%s" % (ci, k)] = v
        print("gdpg config %d %s ok (%d arrays)" % (ci, flags, len(res)))
    names = [str(n) for n in np.load(os.path.join(ROOT, "tests", "golden", "dataset100.npz"))["names"]]
    for ts, nl in TEST_LOOP:
        p = {}
        for fam in ("ER", "BA"):
            p.update(run_test_loop(ts, nl, fam))
        out["test_loop|%s|l%d" % (ts, nl)] = np.array([p[n] for n in names], dtype=np.float64)
        print("mwis_dqn_test.py %s l=%d: mean p = %.6f over %d graphs" % (ts, nl, float(np.mean(list(p.values()))), len(p)))
    np.savez_compressed(OUT, **out)
    print("%s: %d arrays, %d bytes" % (OUT, len(out), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
