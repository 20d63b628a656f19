"""GPU: csrc/big2.hip - the deep c32 stack in one launch for graphs of 977 .. 1 920 vertices (the joint conflict graph of
3 channels x 500 flows, wireless_dqn_test_mc.py:37, 161, 244-289): Z1 in LDS a feature half at a time, every aggregation in
two walks.  Chains per (row, feature) are untouched, so everything must equal the twin bit for bit.

(1) plain solves from 977 to 1 920 vertices - sparse, dense, hubs, ragged batches, biases - against the twin; (2) explicit
input features (the layer-by-layer prelude + k_big2) against the twin's forward; (3) complete iterative searches at 1 500
vertices against the oracle's solvers; (4) the existing any-size suite re-run in a child process with option big2 = 1, which
sends every shape k_big takes down k_big2 instead: each residual step of nine solver variants against the fused kernel,
plain solves against the fused kernel and the twin, hub graphs, 600 .. 976-vertex graphs."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_against_twin(engine, hb, layers):
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    assert engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, layers)
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"])
    assert np.allclose(r["totals"].cpu().numpy(), ref["totals"], rtol=1e-12)


@pytest.mark.parametrize("n,p,count", [(977, 0.01, 3), (1024, 0.02, 2), (1100, 0.004, 3), (1500, 0.01, 3), (1536, 0.003, 2),
                                       (1537, 0.006, 2), (1900, 0.004, 2), (1920, 0.01, 2), (1200, 0.3, 1), (1920, 0.2, 1)])
@pytest.mark.parametrize("num_layer,bias", [(4, False), (3, True), (20, False)])
def test_big2_plain_solve_vs_twin(engine, n, p, count, num_layer, bias):
    """ER graphs of 977 .. 1 920 vertices (eight, twelve and fifteen tiles per wave; dense ones: ER(1 200, 0.3) has 430 000
    entries, ER(1 920, 0.2) 740 000, rows of 400 entries), three depths, with and without biases."""
    from distgcn_amd import datagen
    if num_layer == 20 and p > 0.05:
        pytest.skip("the dense graphs at three depths would be minutes of twin time")
    hb = datagen.er_batch(count, n, p, first_index=70)
    _check_against_twin(engine, hb, datagen.random_model(num_layer, 32, bias=bias, seed=2 + num_layer))


def test_big2_ragged_batch_with_hubs_and_small_graphs(engine):
    """One batch: the joint 3 x 500-flow graph, a 1 900-vertex graph with hubs of 1 000 .. 1 800 entries, a 30-vertex graph, an
    empty one, isolated vertices."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from tests.test_gpu_general import _hub_graph
    rng = np.random.default_rng(31)
    mc = datagen.multichannel_batch(1, 500, 0.03, first_index=5)
    mats = [mc.scipy_graph(0).tocsr(), _hub_graph(1900, [(7, 1800), (1000, 1000), (1899, 1500)], 0.002, rng), sp.csr_matrix((0, 0)),
            _hub_graph(30, [(3, 20)], 0.1, rng), sp.csr_matrix((50, 50))]
    for m in mats:
        m.sort_indices()
    hb = HostBatch.from_scipy(mats, [rng.random(m.shape[0]) for m in mats])
    _check_against_twin(engine, hb, datagen.random_model(5, 32, bias=True, seed=8))


def test_big2_explicit_features_vs_twin_forward(engine):
    """Explicit input features (F = 4): layer index 0 runs layer by layer (transform, aggregation with the row chains in
    double), k_big2 takes over from its output - dgcn_solve_batch with X against the twin's forward and greedy search."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = datagen.er_batch(2, 1400, 0.005, first_index=11)
    layers = datagen.random_model(4, 32, feature_size=4, bias=True, seed=6)
    X = np.random.default_rng(2).uniform(-1, 1, (hb.num_nodes, 4)).astype(np.float32)
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    sc = ctwin.forward(lap, layers, hb.num_nodes, X=X)
    ref = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, sc[:, 0].astype(np.float64) * hb.weights, sum_weights=hb.weights)
    r = engine.solve_fused(db, dm, X=torch.from_numpy(X).to(engine.device))
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), sc.ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])


@pytest.mark.parametrize("which", ["dit", "dit_mis", "cit", "rollout"])
def test_big2_iterative_solvers_vs_oracle(engine, which):
    """Complete searches on a joint 3 x 500-flow graph (1 500 vertices, dit / rollout) and ER(1 100, 0.004) (cit): every step's
    forward is the compaction + k_big2 (predict "mis": weight-derived features, the explicit-feature prelude); decisions equal
    to the oracle's solvers fed with the twin's scores."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from distgcn_amd.runtime_config import FLAGS
    from oracle import ref_numpy as orc
    from tests.test_gpu_general import _twin_scores_fn
    predict = "mis" if which.endswith("_mis") else "mwis"
    agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=3, diver_num=1, max_degree=1, predict=predict), seed=13)
    if which == "cit":
        rng = np.random.default_rng(555)
        ip, ix = datagen.er_graph(1100, 0.004, rng)
        adj = sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(1100, 1100))
        w = rng.random(1100)
    else:
        hb = datagen.multichannel_batch(1, 500, 0.03, first_index=9)
        adj, w = hb.scipy_graph(0).tocsr(), hb.weights.copy()
    kind = which.split("_")[0]
    got = agent.solve_iterative_batch([adj], [w], kind, b=3)
    assert got is not None
    if predict == "mis":
        from distgcn_amd.batch import HostBatch
        from oracle import ctwin

        def fn(adj_nn, wts_nn):  # mwis_gdpg_call.py:88: features = wts_nn / (np.amax(wts_nn) + 1e-9), float32 at the feed
            m = sp.csr_matrix(adj_nn)
            m.sort_indices()
            hb1 = HostBatch.from_csr_lists([m.indptr.astype(np.int64)], [m.indices.astype(np.int64)])
            lap = ctwin.supports(hb1.graph_ptr, hb1.row_ptr, hb1.col_idx)[:3]
            wv = np.asarray(wts_nn, np.float64).reshape(-1, 1)
            return ctwin.forward(lap, agent.model.layers, hb1.num_nodes, X=(wv / (np.amax(wv) + 1e-9)).astype(np.float32))
        want = orc.solve_mwis_dit(fn, adj, w, predict="mis")
    else:
        fn = _twin_scores_fn(agent.model.layers)
        want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[kind](fn, adj, w) if kind != "rollout" else \
            orc.solve_mwis_rollout(fn, adj, w, b=3)
    assert got[0][0] == want[0], which
    assert np.allclose(got[0][1], want[1], rtol=1e-12)


def test_any_size_suite_through_big2(engine):
    """option "big2" = 1 (DGCN_OPTIONS in a child process): every shape k_big takes goes down k_big2 - eight tiles per wave, the
    same code as at 1 500 vertices.  The any-size suite's bit-for-bit checks run unchanged: nine solver variants step by step
    against the fused residual kernel, plain solves against the fused kernel and the twin, graphs of 600 .. 976 vertices, rows
    of 575+ entries."""
    env = dict(os.environ, DGCN_OPTIONS="big2=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_general.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "plain_solve_equals or residual_steps_equal or big_graphs_plain or beyond_575"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_big2_residual_step_in_one_launch_agrees_with_the_compaction_path(engine, tmp_path):
    """A residual step on 977 .. 1 920-vertex graphs is one launch of k_big2<.., RESID> too (the residual graph's support from the
    adjacency and the running state).  Complete dit / cit / rollout searches on three ragged ~1 500-vertex graphs (zero weights
    inside live graphs, biases, leaky last layer) by a child process as built and by one with option big_residual = 0 - the compaction
    launches + k_big2 + k_lgs: same states, step counts, score bits."""
    script = os.path.join(ROOT, "tests", "_wide_witness.py")
    files = {}
    for tag, val in (("one_launch", "1"), ("compaction", "0")):
        files[tag] = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, script, files[tag], "4", "1500"], check=True, env=dict(os.environ, DGCN_OPTIONS="big_residual=" + val), timeout=900)
    a, b = np.load(files["one_launch"]), np.load(files["compaction"])
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith("_scores"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        elif k.endswith("_totals"):
            assert np.allclose(a[k], b[k], rtol=1e-12), k
        else:
            assert np.array_equal(a[k], b[k]), k
    assert a["cit_steps"][0] > 100


@pytest.mark.parametrize("which", ["dit", "cit", "rollout"])
def test_big2_iterative_solvers_restatement_forward(engine, which):
    """Round-5 review item 5: k_big2<RESID> (977 .. 1 920 vertices) held against the oracle's solvers fed with the RESTATEMENT's
    own float32 forward (oracle/ref_numpy._default_scores_fn: the NumPy restatement of gcn/layers.py:189-216,
    mwis_gdpg_call.py:211-216) - one hop from the reference, not two through the twin.  One ER(1 200, 0.005) graph, l = 3,
    complete dit / cit / rollout (b = 4) searches on the device.  (Chosen on the CPU so that no decision lies inside the two
    forwards' rounding distance: the oracle's solvers select the same 440 / 470 / 467 vertices with either forward.)"""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from distgcn_amd.runtime_config import FLAGS
    from oracle import ref_numpy as orc
    n, p = 1200, 0.005
    agent = DQNAgent(FLAGS.copy(feature_size=1, hidden1=32, num_layer=3, diver_num=1, max_degree=1, predict="mwis"), seed=21)
    fn = orc._default_scores_fn(agent.model.layers)
    rng = np.random.default_rng(20230800 + n)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None
    want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else \
        orc.solve_mwis_rollout(fn, adj, w, b=4)
    assert len(want[0]) == {"dit": 440, "cit": 470, "rollout": 467}[which]
    assert got[0][0] == want[0], which
    assert np.allclose(got[0][1], want[1], rtol=1e-12)
