"""GPU: csrc/wide.hip - one-layer models (the reference's multi-channel recipe: bash/twc_major_wireless_mc_test.sh:3
`--num_layer=1 --num_channels=3` on the joint K x nflows graph, wireless_dqn_test_mc.py:161, 244-289) on graphs of any size
up to 9 600 vertices: dgcn_solve_batch in ONE launch, a residual step of solve_mwis_dit / _cit in one, of a rollout in five.

Checks: (1) plain solves against the twin, bit for bit, from 16 to 9 600 vertices, sparse and dense (columns in LDS or
not), with bias / explicit features / F = 16 input features; (2) forced onto shapes the fused residual kernel takes too,
every step of every solver variant must leave the same bytes / bits; (3) complete searches at 900 / 1 500 / 3 000
vertices against the oracle's solvers (oracle/ref_numpy.py, pinned by the executed reference) fed with the twin's scores;
(4) the layer-by-layer any-size path (what ran before) as a second witness via a child process with option wide1 = 0 (DGCN_OPTIONS)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flags(**kw):
    from distgcn_amd.runtime_config import FLAGS
    base = dict(feature_size=1, hidden1=32, num_layer=1, diver_num=1, max_degree=1, predict="mwis")
    base.update(kw)
    return FLAGS.copy(**base)


@pytest.fixture
def general_switch():
    from distgcn_amd import _lib
    lib = _lib.load()
    initial = int(lib.dgcn_get_general())

    def set_to(value):
        lib.dgcn_set_general(-1 if value is None else int(value))
    yield set_to
    lib.dgcn_set_general(initial)


def _twin_scores_fn(layers):
    from distgcn_amd.batch import HostBatch
    from oracle import ctwin
    import scipy.sparse as sp

    def fn(adj_nn, wts_nn):
        a = sp.csr_matrix(adj_nn)
        a.sort_indices()
        hb = HostBatch.from_csr_lists([a.indptr.astype(np.int64)], [a.indices.astype(np.int64)])
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        return ctwin.forward(lap, layers, hb.num_nodes)
    return fn


SHAPES = [(600, 0.01, 3), (900, 0.02, 3), (513, 0.1, 2), (1500, 0.004, 3), (1500, 0.1, 2), (3000, 0.002, 2), (9600, 0.0005, 2),
          (9600, 0.004, 1)]


@pytest.mark.parametrize("n,p,count", SHAPES)
@pytest.mark.parametrize("variant", ["plain", "bias_relu", "features16"])
def test_one_layer_solve_any_size_vs_twin(engine, n, p, count, variant):
    """dgcn_solve_batch, one-layer model, graphs beyond shallow.hip's 512 vertices: scores (bits), sets, rounds, totals
    equal to the twin's; the engine reports the any-size path."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    import torch
    hb = datagen.er_batch(count, n, p, first_index=60)
    F = 16 if variant == "features16" else 1
    layers = datagen.random_model(1, 32, feature_size=F, bias=variant != "plain", seed=3 + F)
    if variant == "bias_relu":
        layers[0]["act"] = "relu"
    X = None
    if variant == "features16":
        X = np.random.default_rng(4).uniform(-1, 1, (hb.num_nodes, F)).astype(np.float32)
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    assert engine.solve_path(db, dm) == 2
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    sc = ctwin.forward(lap, layers, hb.num_nodes, X=X)
    ref = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, sc[:, 0].astype(np.float64) * hb.weights, sum_weights=hb.weights)
    r = engine.solve_fused(db, dm, X=None if X is None else torch.from_numpy(X).to(engine.device))
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), sc.ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"])
    assert np.allclose(r["totals"].cpu().numpy(), ref["totals"], rtol=1e-12)


def test_one_layer_ragged_batch_with_empty_and_isolated(engine):
    """A ragged batch: an empty graph, a single vertex, a graph of isolated vertices, 700- and 2 000-vertex graphs."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(8)
    mats = [sp.csr_matrix((0, 0)), sp.csr_matrix((1, 1)), sp.csr_matrix((40, 40))]
    for n, p in ((700, 0.01), (2000, 0.003)):
        ip, ix = datagen.er_graph(n, p, rng)
        mats.append(sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(n, n)))
    hb = HostBatch.from_scipy(mats, [rng.random(m.shape[0]) for m in mats])
    layers = datagen.random_model(1, 32, seed=1)
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    assert engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, layers)
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"])
    assert r["rounds"].cpu().numpy()[0] == 0 and r["totals"].cpu().numpy()[0] == 0.0


STEPPERS = {  # which -> (greedy mode name, max_rounds, options: scores given / completions by priority, predict)
    "dit": ("GREEDY_ROUNDS", 1, (False, False), "mwis"),
    "lgs_all": ("GREEDY_ROUNDS", 0, (False, False), "mwis"),
    "cit": ("GREEDY_CENTRAL", 1, (False, False), "mwis"),
    "rollout": ("GREEDY_ROLLOUT", 1, (False, False), "mwis"),
    "rollout00": ("GREEDY_ROLLOUT", 1, (True, False), "mwis"),
    "rollout1": ("GREEDY_ROLLOUT", 1, (False, True), "mwis"),
    "dit_mis": ("GREEDY_ROUNDS", 1, (False, False), "mis"),
    "cit_mis": ("GREEDY_CENTRAL", 1, (False, False), "mis"),
    "rollout_mis": ("GREEDY_ROLLOUT", 1, (False, False), "mis"),
}


@pytest.mark.parametrize("which", sorted(STEPPERS))
def test_one_layer_residual_steps_equal_the_fused_kernel(engine, golden, general_switch, which):
    """One call of dgcn_solve_residual_batch = one solver step.  One-layer GCN2_DQN (bias, leaky last layer), from a start with
    decided vertices, a graph with nothing left, a graph without positive weight and zero weights inside a live graph: step by
    step the one-launch kernel of the any-size path must leave the same state bytes, scores (bits), rounds and totals as the
    fused residual kernel, and the host loop must stop after as many steps."""
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    mode_name, max_rounds, (given, by_prio), predict = STEPPERS[which]
    agent = DQNAgent(_flags(num_layer=1, predict=predict), seed=9)
    rng = np.random.default_rng(5)
    for k in agent.model.vars:
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    hb = golden.host_batch([2, 7, 1, 0, 12, 8])
    sl = hb.graph_slices()
    hb.weights[sl[4][0]:sl[4][1]] = 0.0
    hb.weights[sl[2][0]:sl[2][0] + 5] = 0.0
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    init = np.where(rng.random(hb.num_nodes) < 0.2, rng.integers(1, 3, hb.num_nodes), 0).astype(np.uint8)
    init[sl[3][0]:sl[3][1]] = 2
    greedy = getattr(engine, mode_name)
    options = (engine.SCORES_GIVEN if given else 0) | (engine.COMPLETE_BY_PRIORITY if by_prio else 0)
    full_scores = None
    if given:
        general_switch(None)
        full_scores = agent.model.forward_batch(engine, db, X=agent._features(hb), mode=1).clone()
    states, outs = {}, {}
    for path in (None, 1):
        general_switch(path)
        assert engine.solve_path(db, dm) == (1 if path is None else 2)
        states[path] = torch.from_numpy(init.copy()).to(engine.device)
        outs[path] = engine.solve_buffers(db, True)
    steps = 0
    while True:
        snap = {}
        for path in (None, 1):
            general_switch(path)
            res = engine.solve_residual(db, dm, states[path], predict=predict, greedy=greedy, max_rounds=max_rounds, beam=6,
                                        weight_features=predict != "mwis", want_scores=True, max_steps=1, out=outs[path],
                                        options=options, scores=None if full_scores is None else full_scores.clone())
            engine.check_status(res["status"])
            snap[path] = (states[path].cpu().numpy().copy(), outs[path]["rounds"].cpu().numpy().copy(),
                          outs[path]["totals"].cpu().numpy().copy(),
                          None if given else outs[path]["scores"].cpu().numpy().ravel().copy())
        a, b = snap[None], snap[1]
        assert np.array_equal(a[0], b[0]), (which, steps)
        assert np.array_equal(a[1], b[1]), (which, steps, a[1], b[1])
        assert np.allclose(a[2], b[2], rtol=1e-12, atol=0), (which, steps)
        if not given:
            assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)), (which, steps)
        steps += 1
        if not a[1].any():
            break
        assert steps < 400
    assert steps > 1
    st = snap[1][0]
    assert np.array_equal(st[sl[4][0]:sl[4][1]], init[sl[4][0]:sl[4][1]])
    for path in (None, 1):
        general_switch(path)
        s0 = torch.from_numpy(init.copy()).to(engine.device)
        res = engine.solve_residual(db, dm, s0, predict=predict, greedy=greedy, max_rounds=max_rounds, beam=6,
                                    weight_features=predict != "mwis", options=options, finish_small=False,
                                    scores=None if full_scores is None else full_scores.clone())
        assert res["steps"] == steps - 1 and np.array_equal(s0.cpu().numpy(), st), (which, path, res["steps"], steps)


@pytest.mark.parametrize("which,n,p", [("dit", 900, 0.01), ("cit", 900, 0.01), ("rollout", 900, 0.01), ("rollout1", 900, 0.01),
                                       ("dit", 1500, 0.004), ("cit", 1500, 0.004), ("rollout", 1500, 0.004),
                                       ("dit", 3000, 0.002), ("dit", 600, 0.08), ("cit", 600, 0.08), ("rollout1", 600, 0.08)])
def test_one_layer_iterative_solvers_vs_oracle(engine, which, n, p):
    """solve_mwis_dit / _cit / _rollout with a one-layer model on 600 .. 3 000-vertex graphs, entirely on the device:
    decisions equal to the oracle's solvers (oracle/ref_numpy.py, the restated control flow of mwis_gdpg_call.py:278-659)
    fed with the twin's scores (rollouts of 3 000 vertices are left out: a minute of oracle time each).  The forward is two hops from the reference here (twin -> restatement); the restatement's own
    forward feeds the same solvers in test_one_layer_iterative_solvers_restatement_forward."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=1), seed=21)
    fn = _twin_scores_fn(agent.model.layers)
    rng = np.random.default_rng(20231000 + n)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None  # ran through dgcn_solve_residual_batch
    if which == "dit":
        want = orc.solve_mwis_dit(fn, adj, w)
    elif which == "cit":
        want = orc.solve_mwis_cit(fn, adj, w)
    else:
        want = orc.solve_mwis_rollout(fn, adj, w, b=4, by_priority=which == "rollout1")
    assert got[0][0] == want[0], (which, n)
    assert np.allclose(got[0][1], want[1], rtol=1e-12)


@pytest.mark.parametrize("which,n,p", [("dit", 900, 0.01), ("cit", 900, 0.01), ("rollout", 600, 0.08)])
def test_one_layer_iterative_solvers_restatement_forward(engine, which, n, p):
    """The same searches held against the oracle's solvers fed with the RESTATEMENT's own float32 forward
    (oracle/ref_numpy._default_scores_fn: makestate + gcn_forward, the NumPy restatement of gcn/layers.py:189-216) - one hop
    from the reference instead of two.  (No vertex of these graphs is decided inside the two forwards' rounding distance:
    checked on the CPU when the cases were chosen - the oracle's solvers select the same sets with either forward.)"""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=1), seed=21)
    fn = orc._default_scores_fn(agent.model.layers)
    rng = np.random.default_rng(20231000 + n)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None
    want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else \
        orc.solve_mwis_rollout(fn, adj, w, b=4)
    assert got[0][0] == want[0], (which, n)
    assert np.allclose(got[0][1], want[1], rtol=1e-12)


@pytest.mark.parametrize("which", ["dit", "cit", "rollout"])
def test_two_layer_iterative_solvers_restatement_forward(engine, which):
    """Round-5 review item 5: the TWO-layer form of k_wide1 (F -> 32 -> 1 on 900 vertices, every residual step one launch)
    against the oracle's solvers fed with the restatement's own float32 forward - one hop from the reference.  (Chosen on the
    CPU: the oracle's solvers select the same 259 / 275 / 269 vertices with the restatement's and with the twin's forward.)"""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    n, p = 900, 0.01
    agent = DQNAgent(_flags(num_layer=2, hidden1=32), seed=21)
    fn = orc._default_scores_fn(agent.model.layers)
    rng = np.random.default_rng(20230800 + n)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None
    want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else \
        orc.solve_mwis_rollout(fn, adj, w, b=4)
    assert len(want[0]) == {"dit": 259, "cit": 275, "rollout": 269}[which]
    assert got[0][0] == want[0], which
    assert np.allclose(got[0][1], want[1], rtol=1e-12)


def test_wireless_joint_graph_900_one_layer(engine):
    """The multi-channel launcher's own shape and depth (bash/twc_major_wireless_mc_test.sh:3: num_layer=1, num_channels=3):
    the slot loop on two joint 3 x 300 graphs with DGCN-LGS and DGCN-LGS-it against the per-instance restatement."""
    import scipy.sparse as sp
    from distgcn_amd import datagen, wireless
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_dqn_call import DQNAgent as DqnAgent
    from distgcn_amd.mwis_gdpg_call import DQNAgent as GdpgAgent
    from oracle import ctwin, ref_numpy as orc, ref_wireless
    nflows, K, T = 300, 3, 4
    adjs, traffics = [], []
    for i in range(2):
        rng = np.random.default_rng(700 + i)
        indptr, indices = datagen.er_graph(nflows, 0.02, rng)
        base = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(nflows, nflows))
        chans = wireless.multichannel_conflict_simulate(base, k=K, p=0.8, rng=np.random.RandomState(17 + i))
        _, joint = wireless.multichannel_conflict_graph(chans)
        adjs.append(joint)
        traffics.append(wireless.make_traffic(nflows, T, 0.05, n_ch=K, seed=40 + i))
    for algo in ("DGCN-LGS", "DGCN-LGS-it"):
        agent = DqnAgent(1, flags=_flags(num_layer=1)) if algo == "DGCN-LGS" else GdpgAgent(_flags(num_layer=1), seed=4)
        layers = agent.model.layers
        fn = _twin_scores_fn(layers)

        def dgcn_fn(adj, w):  # mwis_dqn_call.py:198-241: prune zero weights, GCN, priority, local greedy, map back
            keep = np.flatnonzero(w > 0)
            if keep.size == 0:
                return set()
            sub = sp.csr_matrix(adj[keep][:, keep])
            sub.sort_indices()
            hb = HostBatch.from_csr_lists([sub.indptr.astype(np.int64)], [sub.indices.astype(np.int64)])
            lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
            prio = ctwin.forward(lap, layers, hb.num_nodes)[:, 0].astype(np.float64) * w[keep]
            st, _ = orc.lgs_vectorised(sub.indptr, sub.indices, prio)
            return set(keep[np.flatnonzero(st == 1)].tolist())

        solver = dgcn_fn if algo == "DGCN-LGS" else (lambda a, w: orc.solve_mwis_dit(fn, a, w)[0] if w.size else set())
        got = wireless.simulate(adjs, traffics, algo=algo, agent=agent, wt_sel="qr")
        for i in range(len(adjs)):
            want = ref_wireless.simulate_one(adjs[i], traffics[i]["arrival_pkts"], traffics[i]["link_rates"], solver, "qr")
            assert np.array_equal(got[i]["queue"], want["queue"]), (algo, i)
            assert np.array_equal(got[i]["depart"], want["depart"]), (algo, i)
            assert got[i]["depart"].sum() > 0


def test_one_layer_paths_agree_with_the_layer_by_layer_path(engine, tmp_path):
    """Second witness: the same plain solve and complete dit / cit / rollout searches (three ragged ~900-vertex graphs, zero
    weights inside live graphs, GCN2_DQN-style bias + leaky last layer) run by a child process as built and by one with
    option wide1 = 0 - the compaction + layer-by-layer + k_lgs chain that served one-layer models beyond 512 vertices before
    wide.hip existed: same states, step counts and score bits."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wide_witness.py")
    files = {}
    for tag, val in (("wide", "1"), ("layered", "0")):
        files[tag] = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, DGCN_OPTIONS="wide1=" + val)
        subprocess.run([sys.executable, script, files[tag]], check=True, env=env, timeout=600)
    a, b = np.load(files["wide"]), np.load(files["layered"])
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith("_scores"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        elif k.endswith("_totals"):
            assert np.allclose(a[k], b[k], rtol=1e-12), k
        else:
            assert np.array_equal(a[k], b[k]), k
    assert a["cit_steps"][0] > 100


# ------------------------------------------------------------------------------------------------ two-layer models (F -> C -> 1)
@pytest.mark.parametrize("n,p,count", [(600, 0.01, 3), (900, 0.02, 2), (513, 0.1, 2), (1500, 0.004, 2), (3000, 0.002, 2), (9600, 0.0005, 1)])
@pytest.mark.parametrize("hidden,bias", [(32, False), (64, True), (5, True), (16, False)])
def test_two_layer_solve_any_size_vs_twin(engine, n, p, count, hidden, bias):
    """The reference launches and ships two-layer models too (num_layer=2 in 13 launcher lines; c32 / c64 / c16 .. l2 checkpoints).
    Beyond 512 vertices they ran layer by layer; on constant input features wide.hip takes them in one launch (H is formed sixteen
    features at a time and fed straight into the second layer's transform: no N x C matrix anywhere).  Scores (bits), sets,
    rounds, totals against the twin; hidden widths 5 .. 64."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = datagen.er_batch(count, n, p, first_index=80)
    layers = datagen.random_model(2, hidden, bias=bias, seed=hidden)
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


@pytest.mark.parametrize("which", ["dit", "lgs_all", "cit", "rollout", "rollout00", "rollout1"])
def test_two_layer_residual_steps_equal_the_fused_kernel(engine, golden, general_switch, which):
    """Two-layer GCN2_DQN (biases, leaky last layer), step by step against the fused residual kernel on fixture graphs - as
    test_one_layer_residual_steps_equal_the_fused_kernel."""
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    mode_name, max_rounds, (given, by_prio), predict = STEPPERS[which]
    agent = DQNAgent(_flags(num_layer=2, predict=predict), seed=19)
    rng = np.random.default_rng(15)
    for k in agent.model.vars:
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    hb = golden.host_batch([2, 7, 1, 0, 12, 8])
    sl = hb.graph_slices()
    hb.weights[sl[4][0]:sl[4][1]] = 0.0
    hb.weights[sl[2][0]:sl[2][0] + 5] = 0.0
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    init = np.where(rng.random(hb.num_nodes) < 0.2, rng.integers(1, 3, hb.num_nodes), 0).astype(np.uint8)
    init[sl[3][0]:sl[3][1]] = 2
    greedy = getattr(engine, mode_name)
    options = (engine.SCORES_GIVEN if given else 0) | (engine.COMPLETE_BY_PRIORITY if by_prio else 0)
    full_scores = None
    if given:
        general_switch(None)
        full_scores = agent.model.forward_batch(engine, db, X=agent._features(hb), mode=1).clone()
    states, outs = {}, {}
    for path in (None, 1):
        general_switch(path)
        assert engine.solve_path(db, dm) == (1 if path is None else 2)
        states[path] = torch.from_numpy(init.copy()).to(engine.device)
        outs[path] = engine.solve_buffers(db, True)
    steps = 0
    while True:
        snap = {}
        for path in (None, 1):
            general_switch(path)
            res = engine.solve_residual(db, dm, states[path], predict=predict, greedy=greedy, max_rounds=max_rounds, beam=6,
                                        want_scores=True, max_steps=1, out=outs[path], options=options,
                                        scores=None if full_scores is None else full_scores.clone())
            engine.check_status(res["status"])
            snap[path] = (states[path].cpu().numpy().copy(), outs[path]["rounds"].cpu().numpy().copy(),
                          outs[path]["totals"].cpu().numpy().copy(),
                          None if given else outs[path]["scores"].cpu().numpy().ravel().copy())
        a, b = snap[None], snap[1]
        assert np.array_equal(a[0], b[0]), (which, steps)
        assert np.array_equal(a[1], b[1]), (which, steps, a[1], b[1])
        assert np.allclose(a[2], b[2], rtol=1e-12, atol=0), (which, steps)
        if not given:
            assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)), (which, steps)
        steps += 1
        if not a[1].any():
            break
        assert steps < 400
    assert steps > 1


@pytest.mark.parametrize("which,n,p", [("dit", 900, 0.01), ("cit", 900, 0.01), ("rollout", 900, 0.01), ("dit", 2000, 0.003)])
def test_two_layer_iterative_solvers_vs_oracle(engine, which, n, p):
    """Complete searches with a two-layer model on 900 / 2 000-vertex graphs against the oracle's solvers fed with the twin's scores."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=2), seed=23)
    fn = _twin_scores_fn(agent.model.layers)
    rng = np.random.default_rng(20231100 + n)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None
    want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else \
        orc.solve_mwis_rollout(fn, adj, w, b=4)
    assert got[0][0] == want[0], (which, n)
    assert np.allclose(got[0][1], want[1], rtol=1e-12)
