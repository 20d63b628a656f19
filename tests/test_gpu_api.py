"""GPU: the drop-in modules (same names, arguments and return types as the reference) against the
oracle and the reference-derived goldens."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M1 = "result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn"
M20 = "result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn"
M16 = "result_IS4SAT_deep_ld1_c16_l4_cheb1_diver1_mwis_dqn"


def _flags(**kw):
    from distgcn_amd.runtime_config import FLAGS
    base = dict(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis")
    base.update(kw)
    return FLAGS.copy(**base)


@pytest.mark.parametrize("variant", ["raw", "ties1", "signed"])
def test_heuristics_dropin(engine, golden, variant):
    from distgcn_amd import heuristics as h
    for i in (0, 1, 8, 12):
        adj = golden.scipy(i).tocoo()  # what a modern loadmat hands out
        k = "g%02d_%s" % (i, variant)
        prio = golden.lgs[k + "_prio"]
        s, tot = h.local_greedy_search(adj, prio)
        assert isinstance(s, set) and all(isinstance(x, int) for x in s) and isinstance(tot, np.float64)
        assert sorted(s) == golden.lgs[k + "_set"].tolist()
        assert tot == pytest.approx(float(golden.lgs[k + "_total"]), rel=1e-12, abs=1e-12)
        s2, _, step = h.local_greedy_search_count(adj, prio.reshape(-1, 1))  # column vectors are flattened
        assert s2 == s and step == int(golden.lgs[k + "_rounds"])
        _, _, step, p2p, bst = h.local_greedy_search_stats(adj, prio)
        assert (step, p2p, bst) == (int(golden.lgs[k + "_rounds"]), int(golden.lgs[k + "_p2p"]), int(golden.lgs[k + "_bst"]))
        *_, oh = h.local_greedy_search_overhead(adj, prio)
        assert oh.dtype == np.float64 and np.array_equal(oh, golden.lgs[k + "_overhead"])
        sn, _, nb = h.local_greedy_search_nstep(adj, prio, nstep=1)
        assert sorted(sn) == golden.lgs[k + "_n1_set"].tolist() and sorted(nb) == golden.lgs[k + "_n1_nb"].tolist()
    adj = golden.scipy(3)
    g, gtot = h.greedy_search(adj, golden.csr(3)[2])
    assert sorted(g) == golden.lgs["g03_raw_greedy_set"].tolist()
    assert gtot == pytest.approx(float(golden.graphs["g03_greedy_utility"]), rel=1e-9)


def test_heuristics_preconditions(engine):
    import scipy.sparse as sp
    from distgcn_amd import heuristics as h
    from distgcn_amd._lib import DgcnError
    a = sp.csr_matrix(np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=float))
    with pytest.raises(DgcnError, match="NaN"):
        h.local_greedy_search(a, [1.0, float("nan"), 2.0])
    loop = sp.csr_matrix(np.array([[1, 1], [1, 0]], dtype=float))
    with pytest.raises(DgcnError, match="self-loop"):
        h.local_greedy_search(loop, [1.0, 2.0])
    assert h.local_greedy_search(sp.csr_matrix((0, 0)), [])[0] == set()
    assert h.local_greedy_search_nstep(a, [3.0, 2.0, 1.0], nstep=0) == (set(), np.float64(0.0), set())


@pytest.mark.parametrize("mname", [M1, M20, M16])
def test_dqn_agent_solve_mwis(engine, golden, mname):
    """mwis_dqn_call.DQNAgent.solve_mwis(adj, wts) -> (set, total, 1.0), incl. zero-weight pruning."""
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ref_numpy as orc
    hid = 16 if "_c16_" in mname else 32
    import re
    nl = int(re.search(r"_l(\d+)_cheb", mname).group(1))
    agent = DQNAgent(1, flags=_flags(hidden1=hid, num_layer=nl))
    agent.model.set_params(golden.params(mname))
    layers = golden.layers(mname)
    rng = np.random.default_rng(1)
    for i in (0, 2, 9):
        adj = golden.scipy(i)
        w = golden.csr(i)[2].copy()
        w[rng.random(w.size) < 0.1] = 0.0  # the pruning branch (mwis_dqn_call.py:202-207)
        got, tot, reward = agent.solve_mwis(adj, w)
        want, wtot, _ = orc.solve_mwis_dqn(layers, adj, w)
        assert got == set(int(x) for x in want) and reward == 1.0
        assert tot == pytest.approx(float(wtot), rel=1e-12)
    res = agent.solve_mwis_batch([golden.scipy(i) for i in (3, 4, 5)], [golden.csr(i)[2] for i in (3, 4, 5)])
    for (got, tot, _), i in zip(res, (3, 4, 5)):
        assert got == agent.solve_mwis(golden.scipy(i), golden.csr(i)[2])[0]
    with pytest.raises(NotImplementedError):
        agent.solve_mwis(golden.scipy(0), golden.csr(0)[2], train=True)


@pytest.mark.parametrize("mname", ["result_IS4SAT_deep_ld1_c1_l1_cheb2_diver1_mwis_dqn",
                                   "result_IS4SAT_deep_ld1_c1_l2_cheb2_diver1_mwis_dqn"])
def test_dqn_agent_chebyshev_order_2(engine, golden, all_models, mname):
    """The two shipped max_degree = 2 checkpoints ([I, L, L.L] supports, three weights per layer,
    gcn/utils.py:268-271 + gcn/layers.py:199-208) through DQNAgent: makestate / predict / solve_mwis."""
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ref_numpy as orc
    meta = all_models.meta(mname)
    assert meta["max_degree"] == 2
    F = meta["feature_size"]
    agent = DQNAgent(F, flags=_flags(feature_size=F, hidden1=meta["hidden"], num_layer=meta["num_layer"], max_degree=2))
    agent.model.set_params(all_models.params(mname))
    layers = all_models.layers(mname)
    assert len(layers[0]["weights"]) == 3
    for i in (0, 2, 9):
        adj, w = golden.scipy(i), golden.csr(i)[2].copy()
        got, tot, _ = agent.solve_mwis(adj, w)
        want, wtot, _ = orc.solve_mwis_dqn(layers, adj, w, feature_size=F, max_degree=2)
        assert got == set(int(x) for x in want) and tot == pytest.approx(float(wtot), rel=1e-12)
    adj, w = golden.scipy(0), golden.csr(0)[2]
    state = agent.makestate(adj, w.reshape(-1, 1))
    act_values, action = agent.predict(state)
    assert np.abs(act_values[:, 0] - all_models.expect(0, mname, "f32")).max() <= 1e-5
    assert len(state["support"]) == 3
    foreign = orc.makestate(adj, w.reshape(-1, 1), F, 2, "dqn_call")  # supports built on the host, as the reference does
    fv, fa = agent.predict(foreign)
    assert np.abs(fv - act_values).max() <= 1e-5 and fa[0] == action[0]
    with pytest.raises(ValueError, match="supports"):
        DQNAgent(F, flags=_flags(feature_size=F, hidden1=meta["hidden"], num_layer=meta["num_layer"])).model.set_params(all_models.params(mname))


def test_predict_state_api(engine, golden):
    """makestate/predict: act_values float32 [N,1], action int64 [1]; a foreign state (supports built
    by the oracle's SciPy code, as the reference would) gives the same scores within 1e-5."""
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(1, flags=_flags())
    agent.model.set_params(golden.params(M20))
    adj, w = golden.scipy(0), golden.csr(0)[2]
    state = agent.makestate(adj, w.reshape(-1, 1))
    act_values, action = agent.predict(state)
    assert act_values.dtype == np.float32 and act_values.shape == (200, 1)
    assert action.dtype == np.int64 and action.shape == (1,) and action[0] == int(np.argmax(act_values[:, 0]))
    assert np.abs(act_values[:, 0] - golden.scores["g00|%s|f64" % M20]).max() <= 1e-5
    assert len(state["support"]) == 2 and state["support"][1][1].dtype == np.float64  # lazily built host copy
    foreign = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "dqn_call")
    fv, fa = agent.predict(foreign)
    assert np.abs(fv - act_values).max() <= 1e-5 and fa[0] == action[0]


def test_gdpg_agent(engine, golden):
    """mwis_gdpg_call.DQNAgent: GCN2_DQN with bias on every layer and leaky_relu on the last."""
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    fl = _flags(num_layer=3)
    agent = DQNAgent(fl, seed=5)
    rng = np.random.default_rng(2)
    for k in agent.model.vars:
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    layers = agent.model.layers
    assert layers[-1]["act"] == "leaky_relu"
    for i in (1, 10):
        adj, w = golden.scipy(i), golden.csr(i)[2]
        got, tot = agent.solve_mwis(adj, w)
        want, wtot = orc.solve_mwis_gdpg(layers, adj, w)
        assert got == set(int(x) for x in want) and tot == pytest.approx(float(wtot), rel=1e-12)
    # predict != 'mwis': features w/(max w + 1e-9), priority = raw score
    agent2 = DQNAgent(_flags(num_layer=3, predict="mis"), seed=5)
    adj, w = golden.scipy(2), golden.csr(2)[2]
    got, tot = agent2.solve_mwis(adj, w)
    want, wtot = orc.solve_mwis_gdpg(agent2.model.layers, adj, w, predict="mis")
    assert got == set(int(x) for x in want)


def test_graph_convolution_layer(engine, golden):
    from distgcn_amd.gcn.layers import GraphConvolution
    from oracle import ctwin
    hb = golden.host_batch([2, 3])
    db = engine.upload(hb)
    layer = GraphConvolution(1, 32, act="leaky_relu", bias=True, seed=4)
    out = layer(engine, db).cpu().numpy()
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    want = ctwin.forward(lap, [layer.layer_dict()], hb.num_nodes)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_sharded_solve_single_rank(engine, golden):
    """parallel.solve_sharded wrapping the HIP engine (world 1 here; world 2 runs on gloo in test_dist_gloo)."""
    from distgcn_amd import datagen, parallel
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = datagen.ba_test2_batch(30)
    layers = datagen.random_model(20, 32)
    dm = DeviceModel(layers, engine.device)

    def local(sub):
        res = engine.solve_fused(engine.upload(sub), dm)
        return {k: res[k].cpu().numpy() for k in ("state", "totals", "rounds")}

    res = parallel.solve_sharded(hb, local)
    ref = ctwin.solve(hb, layers)
    assert np.array_equal(res["state"], ref["state"]) and np.array_equal(res["rounds"], ref["rounds"])


def test_evaluation_harness(engine, golden, tmp_path):
    """A12: the mwis_dqn_test.py loop as one batched call; ratio p = total / greedy_utility pinned by the
    reference-stored greedy_utility and the restatement-derived sets of the fixtures."""
    from distgcn_amd import harness
    from distgcn_amd.mwis_gdpg_call import DQNAgent as GdpgAgent
    from distgcn_amd.mwis_dqn_call import DQNAgent
    agent = DQNAgent(1, flags=_flags())
    agent.model.set_params(golden.params(M20))
    ids = list(range(golden.num_graphs))
    adjs = [golden.scipy(i) for i in ids]
    wts = [golden.csr(i)[2] for i in ids]
    gu = [float(golden.graphs["g%02d_greedy_utility" % i]) for i in ids]
    rows = harness.evaluate(agent, adjs, wts, gu, names=golden.names)
    same = 0
    for i, r in enumerate(rows):
        want_set = golden.scores["g%02d|%s|set" % (i, M20)]
        want_p = wts[i][want_set].sum() / gu[i]
        same += abs(r["p"] - want_p) < 1e-12
        assert 0.7 < r["p"] < 1.4
    assert same >= len(rows) - 1  # float32 reorderings may flip one near-tie
    assert np.allclose(harness.greedy_utilities(adjs, wts), gu, rtol=1e-9)
    rows2 = harness.evaluate(agent, adjs[:3], wts[:3])  # denominators recomputed on the device
    assert [r["p"] for r in rows2] == pytest.approx([r["p"] for r in rows[:3]], rel=1e-9)
    harness.write_csv(rows, str(tmp_path / "out" / "model.csv"))
    assert (tmp_path / "out" / "model.csv").read_text().splitlines()[0] == ",data,p"


def test_evaluation_harness_exploring_branch(engine, golden):
    """mwis_dqn_test.py:244-256 with test=False: with probability epsilon a graph's scores are uniform(0,1) draws.
    The same seeded numpy.random stream replayed on the host (one rand() per graph, uniform(size=n) when it fires) and
    the oracle's greedy search give the expected sets; epsilon = 0 leaves the deterministic path untouched."""
    from distgcn_amd import harness
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(1, flags=_flags())
    agent.model.set_params(golden.params(M20))
    ids = [0, 1, 2, 8, 9, 12]
    adjs = [golden.scipy(i) for i in ids]
    wts = [golden.csr(i)[2] for i in ids]
    gu = [float(golden.graphs["g%02d_greedy_utility" % i]) for i in ids]
    base = harness.evaluate(agent, adjs, wts, gu)
    rs = np.random.RandomState(12)
    rows = harness.evaluate(agent, adjs, wts, gu, epsilon=0.5, rng=rs)
    replay = np.random.RandomState(12)
    fired = 0
    for k, i in enumerate(ids):
        if replay.rand() <= 0.5:
            fired += 1
            act = replay.uniform(size=wts[k].size)
            want, _ = orc.local_greedy_search(adjs[k], act * wts[k])
            assert rows[k]["total"] == pytest.approx(float(wts[k][sorted(want)].sum()), rel=1e-12)
        else:
            assert rows[k]["p"] == pytest.approx(base[k]["p"], rel=1e-12)
    assert 0 < fired < len(ids)
    tiny = harness.evaluate(agent, adjs, wts, gu, epsilon=1e-12, rng=np.random.RandomState(1))
    assert [r["p"] for r in tiny] == pytest.approx([r["p"] for r in base], rel=1e-12)


def test_output_heads_dual_and_skip(engine, golden):
    """GCN2_DQN(is_dual=True) (gcn/models.py:651-653) and GCN_DQN with FLAGS.skip (:505-521): model.outputs against
    the NumPy restatement, through forward_batch, predict and - for the one-output skip model - solve_mwis."""
    from distgcn_amd.gcn.models import GCN2_DQN
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ref_numpy as orc
    hb = golden.host_batch([0, 9, 2])
    db = engine.upload(hb)
    # ---- duelling head: 3 outputs -> 2
    m = GCN2_DQN(None, hidden_dim=32, num_layer=3, bias=True, input_dim=1, output_dim=3, is_dual=True, seed=3)
    out = m.forward_batch(engine, db).cpu().numpy()
    assert out.shape == (hb.num_nodes, 2)
    for gi, (n0, n1) in zip([0, 9, 2], hb.graph_slices()):
        adj, w = golden.scipy(gi), golden.csr(gi)[2]
        state = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "gdpg")
        want, _ = orc.gcn_forward(m.layers, state, np.float32, is_dual=True)
        assert np.abs(out[n0:n1] - want).max() <= 1e-5
    agent_state = {"features": state["features"], "support": state["support"]}
    vals, action = m.predict(agent_state, engine)
    assert vals.shape == (w.size, 2) and action.shape == (2,) and np.array_equal(action, np.argmax(vals, axis=0))
    # ---- skip head: dense over concat([X, last activation])
    agent = DQNAgent(1, flags=_flags(num_layer=3, skip=True), seed=4)
    model = agent.model
    k, b = model.vars["gcn_dqn/dense/kernel"], model.vars["gcn_dqn/dense/bias"]
    assert k.shape == (2, 1) and b.shape == (1,)
    model.vars["gcn_dqn/dense/bias"] = np.array([0.05], np.float32)
    out = model.forward_batch(engine, db).cpu().numpy()
    for gi, (n0, n1) in zip([0, 9, 2], hb.graph_slices()):
        adj, w = golden.scipy(gi), golden.csr(gi)[2]
        state = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "dqn_call")
        last, _ = orc.gcn_forward(model.layers, state, np.float32)
        x = np.full((w.size, 1), 1.0, np.float32)
        want = np.concatenate([x, last], axis=1).astype(np.float32) @ k + np.float32(0.05)
        assert np.abs(out[n0:n1] - want).max() <= 1e-5
    adj, w = golden.scipy(9), golden.csr(9)[2]
    got, tot, _ = agent.solve_mwis(adj, w)
    state = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "dqn_call")
    last, _ = orc.gcn_forward(model.layers, state, np.float32)
    sc = (np.concatenate([np.ones((w.size, 1), np.float32), last], axis=1) @ k + np.float32(0.05))[:, 0]
    want, _ = orc.local_greedy_search(adj, sc.astype(np.float64) * w)
    assert got == set(int(v) for v in want)
    # checkpoint round trip keeps the head
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        model.save(d)
        again = DQNAgent(1, flags=_flags(num_layer=3, skip=True), seed=9)
        again.model.load(d)
        assert np.array_equal(again.model.vars["gcn_dqn/dense/bias"], model.vars["gcn_dqn/dense/bias"])
        assert np.array_equal(again.model.vars["gcn_dqn/dense/kernel"], k)


def test_mixed_size_batch_is_bucketed(engine):
    """A few large graphs among many small ones are solved as two launches (small / large LDS images) with
    identical results."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_dqn_call import DQNAgent, solve_host_batch
    from oracle import ctwin
    rng = np.random.default_rng(4)
    ps, cs, ws = [], [], []
    for n in [60] * 160 + [300] * 40:
        p, c = datagen.er_graph(n, 0.1, rng)
        ps.append(p); cs.append(c); ws.append(rng.random(n))
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    assert len(hb.size_buckets()) == 2
    agent = DQNAgent(1, flags=_flags())
    res = solve_host_batch(engine, agent.model, hb)
    ref = ctwin.solve(hb, agent.model.layers)
    assert np.array_equal(res["state"], ref["state"]) and np.array_equal(res["rounds"], ref["rounds"])
    assert np.array_equal(res["scores"].view(np.uint32), ref["scores"].view(np.uint32))


def _twin_scores_fn(layers):
    """scores_fn for the oracle's iterative solvers computed by the C twin (kernel operation order),
    so the GPU solvers must reproduce the oracle's decisions exactly."""
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


def test_lgs_masked_instances(engine, golden):
    """dgcn_lgs_masked_batch == greedy on the induced subgraph, for several masks at once."""
    import torch
    from oracle import ctwin
    from distgcn_amd.batch import HostBatch
    hb = golden.host_batch([0, 9])
    db = engine.upload(hb)
    rng = np.random.default_rng(3)
    K = 5
    init = (rng.random((K, hb.num_nodes)) < 0.3).astype(np.uint8) * 3
    res = engine.lgs_masked(db, db.weights, torch.from_numpy(init).to(engine.device), K, sum_weights=db.weights)
    engine.check_status(res["status"])
    state = res["state"].cpu().numpy()
    totals = res["totals"].cpu().numpy()
    for k in range(K):
        for g, (n0, n1) in enumerate(hb.graph_slices()):
            keep = np.flatnonzero(init[k, n0:n1] == 0)
            sub = hb.scipy_graph(g)[keep][:, keep].tocsr()
            sub.sort_indices()
            shb = HostBatch.from_csr_lists([sub.indptr.astype(np.int64)], [sub.indices.astype(np.int64)])
            w = hb.weights[n0:n1][keep]
            r = ctwin.lgs(shb.graph_ptr, shb.row_ptr, shb.col_idx, w, sum_weights=w)
            assert np.array_equal(state[k, n0:n1][keep], r["state"])
            assert np.all(state[k, n0:n1][init[k, n0:n1] != 0] == 3)
            assert totals[k, g] == pytest.approx(r["totals"][0], rel=1e-12)


@pytest.mark.parametrize("on_device", [True, False])
@pytest.mark.parametrize("which", ["dit", "cit", "rollout", "cit_wrap", "rollout_wrap", "rollout00", "rollout0", "rollout1"])
def test_iterative_solvers(engine, golden, which, on_device):
    """solve_mwis_dit / _cit / _rollout (+ _wrap) against the oracle restatement, both
    with the residual graph masked on the device and with the host re-slicing fallback."""
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=3), seed=9)
    agent.device_iterative = on_device
    fn = _twin_scores_fn(agent.model.layers)
    for i in ((2, 7) if "rollout" in which else (1, 2, 8)):
        adj, w = golden.scipy(i), golden.csr(i)[2]
        if which == "dit":
            got, want = agent.solve_mwis_dit(adj, w), orc.solve_mwis_dit(fn, adj, w)
        elif which == "cit":
            got, want = agent.solve_mwis_cit(adj, w), orc.solve_mwis_cit(fn, adj, w)
        elif which == "rollout":
            got, want = agent.solve_mwis_rollout(adj, w, b=8), orc.solve_mwis_rollout(fn, adj, w, b=8)
        elif which == "cit_wrap":
            got, want = agent.solve_mwis_cit_wrap(adj, w), orc.solve_wrap(orc.solve_mwis_cit, fn, adj, w)
        elif which in ("rollout00", "rollout0", "rollout1"):  # the variants mwis_gdpg_call.py:413-594
            rescore, by_prio = {"rollout00": (False, False), "rollout0": (False, True), "rollout1": (True, True)}[which]
            got = getattr(agent, "solve_mwis_" + which)(adj, w, b=8)
            want = orc.solve_mwis_rollout(fn, adj, w, b=8, rescore=rescore, by_priority=by_prio)
        else:
            got = agent.solve_mwis_rollout_wrap(adj, w, b=8)
            want = orc.solve_wrap(orc.solve_mwis_rollout, fn, adj, w, b=8)
        assert got[0] == want[0], (which, i)
        assert np.allclose(got[1], want[1], rtol=1e-12)


@pytest.mark.parametrize("which", ["dit", "cit", "rollout"])
def test_residual_solvers_on_device(engine, golden, which):
    """dgcn_solve_residual_batch: the same three solvers with the residual graph masked inside the fused
    kernel (no host re-slicing), a whole batch per launch; decisions must equal the oracle's per graph."""
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=3), seed=9)
    fn = _twin_scores_fn(agent.model.layers)
    ids = [2, 7, 1] if which == "rollout" else [1, 2, 8, 0, 12]
    hb = golden.host_batch(ids)
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    assert engine.solve_supported(db, dm)
    state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
    greedy = {"dit": engine.GREEDY_ROUNDS, "cit": engine.GREEDY_CENTRAL, "rollout": engine.GREEDY_ROLLOUT}[which]
    res = engine.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=8)
    engine.check_status(res["status"])
    st = res["state"].cpu().numpy()
    for g, (i, (n0, n1)) in enumerate(zip(ids, hb.graph_slices())):
        adj, w = golden.scipy(i), golden.csr(i)[2]
        if which == "dit":
            want = orc.solve_mwis_dit(fn, adj, w)
        elif which == "cit":
            want = orc.solve_mwis_cit(fn, adj, w)
        else:
            want = orc.solve_mwis_rollout(fn, adj, w, b=8)
        got = set(int(v) for v in np.flatnonzero(st[n0:n1] == 1))
        assert got == want[0], (which, i)
        assert not np.any(st[n0:n1] == 0)


def test_residual_step_equals_resliced_graph(engine, golden):
    """One masked launch == the plain fused solve of the re-sliced (induced) subgraph, bit for bit."""
    import torch
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=4), seed=4)
    hb = golden.host_batch([0, 3, 9, 12])
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    rng = np.random.default_rng(11)
    gone = rng.random(hb.num_nodes) < 0.35
    n0, n1 = hb.graph_slices()[1]
    gone[n0:n1] = True  # one graph with nothing left
    init = np.where(gone, rng.integers(1, 3, hb.num_nodes), 0).astype(np.uint8)
    state = torch.from_numpy(init.copy()).to(engine.device)
    out = engine.solve_buffers(db, True)
    res = engine.solve_residual(db, dm, state, greedy=engine.GREEDY_ROUNDS, max_rounds=0, want_scores=True,
                                max_steps=1, out=out)
    engine.check_status(res["status"])
    st = res["state"].cpu().numpy()
    sc = res["scores"].cpu().numpy().ravel()
    totals = out["totals"].cpu().numpy()
    assert np.array_equal(st[gone], init[gone])
    ps, cs, ws, keeps = [], [], [], []
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        keep = np.flatnonzero(~gone[n0:n1])
        sub = hb.scipy_graph(g)[keep][:, keep].tocsr()
        sub.sort_indices()
        ps.append(sub.indptr.astype(np.int64)); cs.append(sub.indices.astype(np.int64)); ws.append(hb.weights[n0:n1][keep])
        keeps.append(keep + n0)
    shb = HostBatch.from_csr_lists(ps, cs, ws)
    ref = engine.solve_fused(engine.upload(shb), dm)
    engine.check_status(ref["status"])
    rst, rsc, rtot = ref["state"].cpu().numpy(), ref["scores"].cpu().numpy().ravel(), ref["totals"].cpu().numpy()
    allkeep = np.concatenate(keeps)
    assert np.array_equal(st[allkeep], rst)
    assert np.array_equal(sc[allkeep].view(np.uint32), rsc.view(np.uint32))
    assert np.all(sc[gone] == 0)
    for g in range(hb.num_graphs):
        assert totals[g] == pytest.approx(rtot[g], rel=1e-12)


@pytest.mark.parametrize("which", ["dit", "rollout"])
def test_residual_weight_features(engine, golden, which):
    """predict != 'mwis': features w / (max residual w + 1e-9) (mwis_gdpg_call.py:82-97) are rebuilt per step
    inside the kernel (feature_mode 1); decisions equal the host re-slicing path and the oracle."""
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=3, predict="mis"), seed=5)
    fn = orc._default_scores_fn(agent.model.layers, predict="mis")
    for i in (1, 8):
        adj, w = golden.scipy(i), golden.csr(i)[2]
        res = {}
        for dev in (True, False):
            agent.device_iterative = dev
            res[dev] = agent.solve_mwis_dit(adj, w) if which == "dit" else agent.solve_mwis_rollout(adj, w, b=4)
        assert res[True][0] == res[False][0]
        assert np.allclose(res[True][1], res[False][1], rtol=1e-12)
        if which == "dit":
            want = orc.solve_mwis_dit(fn, adj, w, predict="mis")
            # the float64 oracle may order near-equal scores differently; sets agree on these fixtures
            assert res[True][0] == want[0]


@pytest.mark.parametrize("family", ["er", "ba"])
def test_c5_rollout_n500(engine, family):
    """BASELINE config 5: GCN-guided tree search + 1-step rollout (b=16) on N=500 conflict graphs
    (stand-ins ER G(500, 0.02) and BA(500, 5)), solved on the device, against the oracle."""
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    rng = np.random.default_rng(20230605)
    if family == "er":
        indptr, indices = datagen.er_graph(500, 0.02, rng)
    else:
        indptr, indices = datagen.ba_graph(500, 5, rng)
    w = rng.random(500)
    import scipy.sparse as sp
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(500, 500))
    agent = DQNAgent(_flags(num_layer=3), seed=21)
    fn = _twin_scores_fn(agent.model.layers)
    got = agent.solve_iterative_batch([adj], [w], "rollout", b=16)
    assert got is not None  # ran through dgcn_solve_residual_batch
    want = orc.solve_mwis_rollout(fn, adj, w, b=16)
    assert got[0][0] == want[0]
    assert np.allclose(got[0][1], want[1], rtol=1e-12)
    # independent set, maximal
    sel = np.zeros(500, bool); sel[list(got[0][0])] = True
    assert not (adj[sel][:, sel]).nnz
    assert np.all((adj @ sel.astype(np.float64) > 0) | sel)


def test_c5_rollout_full_size_trained_l20(engine, golden):
    """BASELINE config 5 at full size: 64 ER G(500, 0.02) conflict graphs, the trained IS4SAT l=20 c=32 weights,
    rollout search b=16, all 64 graphs advanced together on the device (one launch per step).  The Python oracle
    needs minutes per graph at l=20, so the check is: independence + maximality on all 64, and - for three of
    them - the same set as the host-re-slicing path (the reference's own control flow: SciPy re-slicing per step,
    mwis_gdpg_call.py:596-659, with every forward pass and greedy completion on the device)."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.gcn.models import GCN_DQN
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=20), seed=1)
    agent.model = GCN_DQN(None, input_dim=1, flags=_flags(num_layer=20))
    agent.model.set_params(golden.params(M20))
    adjs, wts = [], []
    for g in range(64):
        rng = np.random.default_rng(20230700 + g)
        indptr, indices = datagen.er_graph(500, 0.02, rng)
        adjs.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(500, 500)))
        wts.append(rng.random(500))
    got = agent.solve_iterative_batch(adjs, wts, "rollout", b=16)
    assert got is not None and len(got) == 64
    for (sel, total), adj, w in zip(got, adjs, wts):
        m = np.zeros(500, bool); m[list(sel)] = True
        assert not (adj[m][:, m]).nnz and np.all((adj @ m.astype(np.float64) > 0) | m)
        assert float(np.asarray(total).ravel()[0]) == pytest.approx(float(w[m].sum()), rel=1e-12)
    agent.device_iterative = False
    for g in (0, 31, 63):
        sel, total = agent.solve_mwis_rollout(adjs[g], wts[g], b=16)
        assert sel == got[g][0], g


def test_large_graph_takes_layered_path(engine):
    """Graphs beyond the fused kernel's 512 vertices / 160 KB image run layer by layer, same results."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.mwis_dqn_call import DQNAgent, solve_host_batch
    from oracle import ctwin
    for n, p, fused in ((600, 0.01, False), (500, 0.1, False), (500, 0.02, True)):
        hb = datagen.er_batch(3, n, p)
        agent = DQNAgent(1, flags=_flags(num_layer=4))
        # N=500 with ~5 000 edges fits the fused kernel with its entry values in global scratch
        assert engine.solve_supported(engine.upload(hb), DeviceModel(agent.model.layers, engine.device)) == fused
        res = solve_host_batch(engine, agent.model, hb)
        ref = ctwin.solve(hb, agent.model.layers)
        assert np.array_equal(res["state"], ref["state"])
        assert np.array_equal(res["scores"].view(np.uint32), ref["scores"].view(np.uint32))


def test_agent_helper_methods(engine, golden):
    """utility / topology_encode / schedule / solve_mwis_util / mellowmax (mwis_gdpg_call.py:140-276)."""
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=3), seed=9)
    adj, w = golden.scipy(2), golden.csr(2)[2]
    act_vals, state = agent.utility(adj, w)
    assert act_vals.shape == (w.size, 1) and act_vals.dtype == np.float32
    assert np.array_equal(agent.topology_encode(adj, w), act_vals)
    mwis, total, _, av = agent.schedule(adj, w)
    ref_set, ref_total = agent.solve_mwis(adj, w)
    assert mwis == ref_set and total == ref_total and np.array_equal(av, act_vals)
    wu = np.arange(w.size, dtype=np.float64)
    s2, t2 = agent.solve_mwis_util(adj, w, wu)
    assert s2 == ref_set and t2 == pytest.approx(sum(wu[list(ref_set)]))
    q = np.array([0.1, 0.7, 0.3])
    c = q.max()
    assert agent.mellowmax(q, 5.0) == pytest.approx(c + np.log(np.mean(np.exp(5.0 * (q - c)))) / 5.0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("force_dist", [False, True])
def test_bench_contract(force_dist):
    """bench.py prints ONE JSON line (the last line of stdout) with the driver's keys, the roofline and
    cpu_baseline objects; with DGCN_BENCH_FORCE_DIST=1 the RCCL gather path runs on one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if force_dist:
        env.update(DGCN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "1",
                        "--no-cpu-pool"], env=env, capture_output=True, text=True, timeout=500, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["unit"] == "graphs/s"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    # (the C3 launch never leaves the chip: what paces k_fused is the LDS array + the fp32 MFMA pipe; `achieved` stays the
    # SURVEY 8d equivalent bandwidth against the HBM peak - round-4 review, hygiene item)
    assert rf["bound"] == "hbm" and rf["paced_by"] == "lds+mfma" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"  # (frac belongs to `bound`)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and 0.05 < rf["frac"] < 1.0
    assert d["value"] == pytest.approx(500 * 5 / (d["ms_per_step"] * 5e-3), rel=1e-6)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert d["value"] > 10 * cb["value"]  # the north star's >= 10x CPU reference


@pytest.mark.parametrize("algo", ["Greedy", "DGCN-LGS"])
def test_wireless_simulation_matches_restatement(engine, algo):
    """the slot loop of wireless_dqn_test.py:219-293, all instances in lockstep on the device,
    against the per-instance CPU restatement (oracle/ref_wireless.py) with the twin as its scheduler."""
    import scipy.sparse as sp
    from distgcn_amd import datagen, wireless
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from oracle import ctwin, ref_numpy as orc, ref_wireless
    rng = np.random.default_rng(77)
    agent = DQNAgent(1, flags=_flags(num_layer=3))
    layers = agent.model.layers
    adjs, traffics = [], []
    for i, (nflows, n_ch, p) in enumerate([(30, 1, 0.15), (45, 1, 0.08), (20, 2, 0.1)]):
        indptr, indices = datagen.er_graph(nflows * n_ch, p, rng)
        adjs.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(nflows * n_ch,) * 2))
        traffics.append(wireless.make_traffic(nflows, 25, 0.03, n_ch=n_ch, seed=i))

    def greedy_fn(adj, w):
        st, _ = orc.lgs_vectorised(adj.indptr, adj.indices, w)
        return set(np.flatnonzero(st == 1).tolist())

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

    got = wireless.simulate(adjs, traffics, algo=algo, agent=agent, wt_sel="qr")
    for i in range(len(adjs)):
        want = ref_wireless.simulate_one(adjs[i], traffics[i]["arrival_pkts"], traffics[i]["link_rates"],
                                         greedy_fn if algo == "Greedy" else dgcn_fn, "qr")
        assert np.array_equal(got[i]["queue"], want["queue"]), i
        assert np.array_equal(got[i]["depart"], want["depart"]), i
        assert np.allclose(got[i]["total_wt"], want["total_wt"], rtol=1e-12)
        assert got[i]["queue"].max() > 0 and got[i]["depart"].sum() > 0
    assert set(wireless.summarize(got[0])) == {"avg_queue_len", "50p_queue_len", "95p_queue_len", "5p_queue_len"}


@pytest.mark.parametrize("algo", ["DGCN-LGS-it", "DGCN-RS"])
def test_wireless_simulation_iterative_schedulers(engine, algo):
    """The slot loop with the reference's iterative schedulers: 'DGCN-LGS-it' = solve_mwis_dit
    (wireless_dqn_test.py:251-254) and 'DGCN-RS' = solve_mwis_rollout_wrap (:256-260, per connected component), all
    instances (components) advanced by the same launches, against the per-instance CPU restatement with the oracle's
    solvers (pinned by the executed reference) fed by the twin's scores."""
    import scipy.sparse as sp
    from distgcn_amd import datagen, wireless
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc, ref_wireless
    rng = np.random.default_rng(78)
    agent = DQNAgent(_flags(num_layer=3), seed=4)
    for k in agent.model.vars:  # non-zero biases
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    fn = _twin_scores_fn(agent.model.layers)
    adjs, traffics = [], []
    for i, (nflows, n_ch, p) in enumerate([(30, 1, 0.06), (40, 1, 0.04), (18, 2, 0.08)]):  # sparse: several components
        indptr, indices = datagen.er_graph(nflows * n_ch, p, rng)
        adjs.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(nflows * n_ch,) * 2))
        traffics.append(wireless.make_traffic(nflows, 12, 0.03, n_ch=n_ch, seed=10 + i))

    def dit_fn(adj, w):
        return orc.solve_mwis_dit(fn, adj, w)[0] if w.size else set()

    def rs_fn(adj, w):
        return orc.solve_wrap(orc.solve_mwis_rollout, fn, adj, w, b=16)[0] if w.size else set()

    got = wireless.simulate(adjs, traffics, algo=algo, agent=agent, wt_sel="qr")
    for i in range(len(adjs)):
        want = ref_wireless.simulate_one(adjs[i], traffics[i]["arrival_pkts"], traffics[i]["link_rates"],
                                         dit_fn if algo == "DGCN-LGS-it" else rs_fn, "qr")
        assert np.array_equal(got[i]["queue"], want["queue"]), i
        assert np.array_equal(got[i]["depart"], want["depart"]), i
        assert np.allclose(got[i]["total_wt"], want["total_wt"], rtol=1e-12)
        assert got[i]["depart"].sum() > 0
    # the per-component wrappers of the agent batch their components into one device call: same result as the oracle
    adj, w = adjs[1], rng.random(adjs[1].shape[0])
    for name, inner in (("solve_mwis_cit_wrap", orc.solve_mwis_cit), ("solve_mwis_rollout_wrap", orc.solve_mwis_rollout)):
        sol, tot = getattr(agent, name)(adj, w)
        want, wtot = orc.solve_wrap(inner, fn, adj, w)
        assert sol == want and float(np.asarray(tot).ravel()[0]) == pytest.approx(float(wtot[0]), rel=1e-12), name


@pytest.mark.parametrize("algo", ["LGS-Seq", "DGCN-LGS-Seq", "CGCN-RS-Seq"])
def test_wireless_multichannel_sequential_schedulers(engine, algo):
    """wireless_dqn_test_mc.py:292-354 (--opt 5/6/7): channels scheduled one after the other on their own conflict graphs,
    queue estimates handed from channel to channel; all instances' calls of one (slot, channel) batched; against the
    per-instance CPU restatement (oracle/ref_wireless.simulate_seq_one) with the oracle's solvers on the twin's scores."""
    import scipy.sparse as sp
    from distgcn_amd import datagen, wireless
    from distgcn_amd.batch import HostBatch
    from oracle import ctwin, ref_numpy as orc, ref_wireless
    rng = np.random.default_rng(79)
    if algo == "CGCN-RS-Seq":
        from distgcn_amd.mwis_gdpg_call import DQNAgent
        agent = DQNAgent(_flags(num_layer=3), seed=5)
    else:
        from distgcn_amd.mwis_dqn_call import DQNAgent
        agent = DQNAgent(1, flags=_flags(num_layer=3))
    layers = agent.model.layers
    fn = _twin_scores_fn(layers)
    adj_lists, traffics = [], []
    for i, (nflows, n_ch, p) in enumerate([(30, 3, 0.1), (24, 1, 0.12), (40, 2, 0.05)]):
        al = []
        for c in range(n_ch):
            indptr, indices = datagen.er_graph(nflows, p, rng)
            al.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(nflows, nflows)))
        adj_lists.append(al)
        traffics.append(wireless.make_traffic(nflows, 14, 0.05, n_ch=n_ch, seed=20 + i))

    def greedy_fn(adj, w):
        adj = sp.csr_matrix(adj); adj.sort_indices()
        st, _ = orc.lgs_vectorised(adj.indptr, adj.indices, w)
        return set(np.flatnonzero(st == 1).tolist())

    def dgcn_fn(adj, w):  # all weights non-zero here: GCN, priority, local greedy
        sub = sp.csr_matrix(adj); sub.sort_indices()
        hb = HostBatch.from_csr_lists([sub.indptr.astype(np.int64)], [sub.indices.astype(np.int64)])
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        prio = ctwin.forward(lap, layers, hb.num_nodes)[:, 0].astype(np.float64) * w
        st, _ = orc.lgs_vectorised(sub.indptr, sub.indices, prio)
        return set(np.flatnonzero(st == 1).tolist())

    def rs_fn(adj, w):
        return orc.solve_wrap(orc.solve_mwis_rollout, fn, sp.csr_matrix(adj), w, b=16)[0]

    solve = {"LGS-Seq": greedy_fn, "DGCN-LGS-Seq": dgcn_fn, "CGCN-RS-Seq": rs_fn}[algo]
    got = wireless.simulate_seq(adj_lists, traffics, algo=algo, agent=agent)
    multi = 0
    for i in range(len(adj_lists)):
        want = ref_wireless.simulate_seq_one(adj_lists[i], traffics[i]["arrival_pkts"], traffics[i]["link_rates"], solve)
        assert np.array_equal(got[i]["queue"], want["queue"]), i
        assert np.array_equal(got[i]["depart"], want["depart"]), i
        assert got[i]["depart"].sum() > 0
        multi += int(got[i]["scheduled"].max() > traffics[i]["arrival_pkts"].shape[1] // 4)
    assert multi > 0
    with pytest.raises(ValueError, match="shapes disagree"):
        wireless.simulate_seq([adj_lists[0][:2]], traffics[:1], algo=algo, agent=agent)


def _example_inputs():
    """The inputs examples/solve_batch.cpp builds (same 64-bit LCG stream)."""
    from distgcn_amd.batch import HostBatch
    state = [12345]

    def nxt():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return (state[0] >> 40) / 16777216.0

    ps, cs, ws = [], [], []
    for n, dens in ((40, 0.15), (25, 0.3)):
        adj = [[] for _ in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                if nxt() < dens:
                    adj[i].append(j)
                    adj[j].append(i)
        ps.append(np.concatenate([[0], np.cumsum([len(a) for a in adj])]).astype(np.int64))
        cs.append(np.array([j for a in adj for j in a], dtype=np.int64))
        ws.append(np.array([0.05 + nxt() for _ in range(n)]))
    dims = [1, 32, 32, 1]
    layers = []
    for l in range(3):
        lim = np.sqrt(6.0 / (dims[l] + dims[l + 1]))
        flat = np.array([np.float32((2.0 * nxt() - 1.0) * lim) for _ in range(dims[l] * 2 * dims[l + 1])], dtype=np.float32)
        cat = flat.reshape(dims[l], 2 * dims[l + 1])
        layers.append({"weights": [cat[:, :dims[l + 1]].copy(), cat[:, dims[l + 1]:].copy()], "bias": None,
                       "act": "linear" if l == 2 else "leaky_relu"})
    return HostBatch.from_csr_lists(ps, cs, ws), layers


def test_native_cpp_caller_of_the_c_abi():
    """examples/solve_batch (C++, HIP runtime + include/dgcn.h, no Python / torch in the process) against the twin."""
    import subprocess
    from oracle import ctwin
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "solve_batch")
    if not os.path.isfile(exe):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    hb, layers = _example_inputs()
    ref = ctwin.solve(hb, layers)
    assert lines[0].split() == ["dgcn", "100"]
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        tok = lines[1 + g].split()
        assert tok[:2] == ["graph", str(g)] and int(tok[3]) == int(ref["rounds"][g])
        assert float(tok[5]) == pytest.approx(ref["totals"][g], rel=1e-12)
        assert [int(x) for x in tok[7:]] == np.flatnonzero(ref["state"][n0:n1] == 1).tolist()
    scores = np.array([float(x) for x in lines[3].split()[1:]], dtype=np.float32)
    assert np.array_equal(scores.view(np.uint32), ref["scores"][:, 0].view(np.uint32))


def test_known_answers_of_100_shipped_graphs_on_gpu(engine, dataset100, all_models):
    """The reference's stored greedy_utility and its local_greedy_search round counts on 100 shipped graphs, and
    the approximation ratios p = total / greedy_utility (mwis_dqn_test.py:321) of four shipped checkpoints."""
    z = dataset100.z
    hb = dataset100.host_batch()
    db = engine.upload(hb)
    res = engine.lgs(db, prio=db.weights, sum_weights=db.weights)
    engine.check_status(res["status"])
    assert np.allclose(res["totals"].cpu().numpy(), z["greedy_utility"], rtol=1e-9, atol=0)
    assert np.array_equal(res["rounds"].cpu().numpy(), z["lgs_rounds"])
    from distgcn_amd.engine import DeviceModel
    for key in [k for k in z.files if k.startswith("ratio|")]:
        name = key.split("|", 1)[1]
        dm = DeviceModel(all_models.layers(name), engine.device)
        out = engine.fetch_solve_buffers(_solve(engine, db, dm), hb.num_nodes, hb.num_graphs)
        assert out["status"] == 0
        p = out["totals"] / z["greedy_utility"]
        want = z[key]
        # every one of the 100 ratios: the kernels' last-bit differences from the restatement flip no set here
        assert np.sum(~np.isclose(p, want, rtol=1e-9)) == 0, (name, np.flatnonzero(~np.isclose(p, want, rtol=1e-9)).tolist())
        assert 0.75 < p.min() and p.max() < 1.5


def test_test_loop_against_the_executed_reference(engine, dataset100, all_models):
    """A12 against the reference's own run of mwis_dqn_test.py (tests/golden/ref_exec.npz test_loop|*): harness.evaluate on
    the same 100 shipped graphs with the same four checkpoints gives the reference's ratio column, every row of it."""
    import scipy.sparse as sp
    from distgcn_amd import harness
    from distgcn_amd.mwis_dqn_call import DQNAgent
    z = _ref_exec()[0]
    adjs, wts = [], []
    for i in range(dataset100.n):
        p, c, w = dataset100.csr(i)
        adjs.append(sp.csr_matrix((np.ones(c.size), c, p), shape=(w.size, w.size)))
        wts.append(w)
    gu = [float(x) for x in dataset100.z["greedy_utility"]]
    for ts, nl in (("IS4SAT", 1), ("IS4SAT", 20), ("DQNBA", 1), ("DQNBA", 20)):
        name = "result_%s_deep_ld1_c32_l%d_cheb1_diver1_mwis_dqn" % (ts, nl)
        agent = DQNAgent(1, flags=_flags(num_layer=nl))
        agent.model.set_params(all_models.params(name))
        rows = harness.evaluate(agent, adjs, wts, gu)
        p = np.array([r["p"] for r in rows])
        ref = z["test_loop|%s|l%d" % (ts, nl)]
        assert np.sum(~np.isclose(p, ref, rtol=1e-9)) == 0, (name, np.flatnonzero(~np.isclose(p, ref, rtol=1e-9)).tolist())


def _solve(engine, db, dm):
    out = engine.solve_buffers(db, False)
    engine.solve_fused(db, dm, want_scores=False, out=out)
    return out


def test_serving_pipeline_host_to_host(engine, golden):
    """distgcn_amd.serving.SolvePipeline: per-graph CSR arrays in host memory -> native packing into pinned memory ->
    one copy in, one fused launch, one copy out; several batches in flight, results in submission order and equal to
    the twin's; a slot must be read before it is re-used; faults surface at result()."""
    from distgcn_amd import datagen
    from distgcn_amd._lib import DgcnError
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.serving import SolvePipeline
    from oracle import ctwin
    layers = golden.layers(M20)
    dm = DeviceModel(layers, engine.device)
    pipe = SolvePipeline(engine, dm, depth=3)
    batches, refs = [], []
    for k, (count, n) in enumerate([(40, 200), (7, 60), (64, 120), (1, 300), (33, 200)]):
        hb = datagen.er_batch(count, n, 0.1, first_index=1000 * k)
        ps, cs, ws = [], [], []
        for n0, n1 in hb.graph_slices():
            e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
            ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0)); cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0))
            ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
        batches.append((ps, cs, ws))
        refs.append(ctwin.solve(hb, layers))
    got = list(pipe.solve_many(batches))
    assert len(got) == len(refs)
    for g, r in zip(got, refs):
        assert np.array_equal(g["state"], r["state"]) and np.array_equal(g["rounds"], r["rounds"])
        assert np.allclose(g["totals"], r["totals"], rtol=1e-12, atol=0)
    slots = [pipe.submit(*batches[0]) for _ in range(3)]
    with pytest.raises(RuntimeError, match="unread result"):
        pipe.submit(*batches[0])
    for s_ in slots:
        assert np.array_equal(pipe.result(s_)["state"], refs[0]["state"])
    bad = (batches[1][0], batches[1][1], [w.copy() for w in batches[1][2]])
    bad[2][3][0] = np.nan
    with pytest.raises(DgcnError, match="NaN"):
        pipe.result(pipe.submit(*bad))
    assert np.array_equal(pipe.result(pipe.submit(*batches[1]))["state"], refs[1]["state"])  # the pipeline recovers


def test_native_host_solver(engine, golden):
    """dgcn_host_solver_* (csrc/host_solver.hip) through distgcn_amd.serving.HostSolver: the whole host-to-host call in
    native code.  Same sets / rounds / totals / scores as the twin for ragged batches and for single graphs, two batches
    in flight, a slot must be read before re-use and cannot be read twice, data faults surface at result() and the object
    recovers, shapes outside the fused kernel take the any-size path, empty graphs are fine."""
    from distgcn_amd import datagen
    from distgcn_amd._lib import DgcnError
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.serving import HostSolver
    from oracle import ctwin
    layers = golden.layers(M20)
    dm = DeviceModel(layers, engine.device)

    def lists(hb):
        ps, cs, ws = [], [], []
        for n0, n1 in hb.graph_slices():
            e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
            ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0)); cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0))
            ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
        return ps, cs, ws
    hs = HostSolver(engine, dm, depth=2, want_scores=True)
    batches, refs = [], []
    for k, (count, n) in enumerate([(40, 200), (1, 200), (7, 60), (1, 1), (33, 120)]):
        hb = datagen.er_batch(count, n, 0.1, first_index=3000 + 100 * k)
        batches.append(lists(hb))
        refs.append(ctwin.solve(hb, layers))
    for b, r in zip(batches, refs):
        g = hs.solve(*b)
        assert np.array_equal(g["state"], r["state"]) and np.array_equal(g["rounds"], r["rounds"])
        assert np.allclose(g["totals"], r["totals"], rtol=1e-12, atol=0)
        assert np.array_equal(g["scores"].view(np.uint32), np.asarray(r["scores"], np.float32).ravel().view(np.uint32))
    s0, s1 = hs.submit(*batches[0]), hs.submit(*batches[4])  # two in flight
    with pytest.raises(DgcnError, match="unread result"):
        hs.submit(*batches[1])
    assert np.array_equal(hs.result(s1)["state"], refs[4]["state"]) and np.array_equal(hs.result(s0)["state"], refs[0]["state"])
    with pytest.raises(DgcnError, match="holds no result"):
        hs.result(s0)
    bad = (batches[2][0], batches[2][1], [w.copy() for w in batches[2][2]])
    bad[2][3][0] = np.nan
    with pytest.raises(DgcnError, match="NaN"):
        hs.solve(*bad)
    assert np.array_equal(hs.solve(*batches[2])["state"], refs[2]["state"])  # recovers
    big = datagen.er_batch(1, 600, 0.01, first_index=77)  # beyond the fused kernel: dgcn_solve_batch's any-size path, same calls
    gb, rb = hs.solve(*lists(big)), ctwin.solve(big, layers)
    assert np.array_equal(gb["state"], rb["state"]) and np.array_equal(gb["rounds"], rb["rounds"])
    assert np.array_equal(gb["scores"].view(np.uint32), np.asarray(rb["scores"], np.float32).ravel().view(np.uint32))
    empty = (np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    g = hs.solve([empty[0], batches[3][0][0]], [empty[1], batches[3][1][0]], [empty[2], batches[3][2][0]])
    assert g["state"].size == 1 and g["totals"].size == 2 and g["totals"][0] == 0.0 and g["rounds"][0] == 0
    g = hs.solve([empty[0]], [empty[1]], [empty[2]])
    assert g["state"].size == 0 and g["totals"].tolist() == [0.0] and g["rounds"].tolist() == [0]
    with pytest.raises(ValueError, match="does not match"):
        hs.submit([batches[1][0][0]], [batches[1][1][0][:-2]], [batches[1][2][0]])
    with pytest.raises(ValueError, match="does not match"):  # the one-call path of the CPython helper checks the same
        hs.solve([batches[1][0][0]], [batches[1][1][0][:-2]], [batches[1][2][0]])
    with pytest.raises(TypeError):
        hs.solve([batches[1][0][0].astype(np.float32)], [batches[1][1][0]], [batches[1][2][0]])
    # small batches are read in place from the pinned staging memory and write their results there; larger ones (or
    # option host_direct_bytes = 0) take the two copies: same answers, through both the one-call and the two-call path
    import os
    from distgcn_amd import _lib
    for direct in (0, -1):
        _lib.set_option("host_direct_bytes", direct)
        for b, r in zip(batches, refs):
            for g in (hs.solve(*b), hs.result(hs.submit(*b))):
                assert np.array_equal(g["state"], r["state"]) and np.array_equal(g["rounds"], r["rounds"])
                assert np.allclose(g["totals"], r["totals"], rtol=1e-12, atol=0)
                assert np.array_equal(g["scores"].view(np.uint32), np.asarray(r["scores"], np.float32).ravel().view(np.uint32))
    hs.close()
    # The one-slot object's kernel tells the host itself when a batch is through (a word in pinned memory, written after the
    # outputs have been written back; option host_done_word = 0 = wait for the event only): 400 single-graph calls in a row,
    # every result complete when it is handed out.
    one = HostSolver(engine, dm, depth=1, want_scores=True)
    ps, cs, ws = batches[0]
    gp = np.concatenate([[0], np.cumsum([p.size - 1 for p in ps])])
    sc0 = np.asarray(refs[0]["scores"], np.float32).ravel()
    for i in range(400):
        k = (7 * i) % len(ps)
        g = one.solve([ps[k]], [cs[k]], [ws[k]])
        assert np.array_equal(g["state"], refs[0]["state"][gp[k]:gp[k + 1]]) and g["rounds"][0] == refs[0]["rounds"][k], i
        assert np.array_equal(g["scores"].view(np.uint32), sc0[gp[k]:gp[k + 1]].view(np.uint32)), i
        assert abs(g["totals"][0] - refs[0]["totals"][k]) <= 1e-12 * abs(refs[0]["totals"][k]), i
    # the paths that leave the kernel early count themselves too: an empty graph beside a real one, a NaN weight
    g = one.solve([empty[0], ps[3]], [empty[1], cs[3]], [empty[2], ws[3]])
    assert g["totals"][0] == 0.0 and g["rounds"][0] == 0 and np.array_equal(g["state"], refs[0]["state"][gp[3]:gp[4]])
    wbad = ws[5].copy(); wbad[0] = np.nan
    with pytest.raises(DgcnError, match="NaN"):
        one.solve([ps[5]], [cs[5]], [wbad])
    for k in (5, 6, 5):
        g = one.solve([ps[k]], [cs[k]], [ws[k]])
        assert np.array_equal(g["state"], refs[0]["state"][gp[k]:gp[k + 1]])
    one.close()
    # A placement fault of the several-workgroups-per-graph kernel (injected: option test_cluster_fault) is not the
    # caller's problem: the object switches the variant off for the process, solves the batch again and hands out that.
    lib = _lib.load()
    initial = int(lib.dgcn_get_cluster())
    _lib.set_option("test_cluster_fault", 1)
    try:
        for direct in (0, -1):
            _lib.set_option("host_direct_bytes", direct)
            lib.dgcn_set_cluster(-1)
            one = HostSolver(engine, dm, depth=1, want_scores=True)
            g = one.solve(*batches[1])  # one N = 200 graph, 20 layers: the cluster variant's case
            assert int(lib.dgcn_get_cluster()) == 0  # switched off process-wide by the library itself (an atomic, not setenv)
            assert np.array_equal(g["state"], refs[1]["state"]) and np.array_equal(g["rounds"], refs[1]["rounds"])
            assert np.array_equal(g["scores"].view(np.uint32), np.asarray(refs[1]["scores"], np.float32).ravel().view(np.uint32))
            one.close()
    finally:
        _lib.set_option("test_cluster_fault", 0)
        _lib.set_option("host_direct_bytes", -1)
        lib.dgcn_set_cluster(initial)


def _ref_exec():
    import json
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "ref_exec.npz"))
    dqn = [(str(m), json.loads(str(f))) for m, f in zip(z["dqn_models"], z["dqn_flags"])]
    gdpg = [json.loads(str(f)) for f in z["gdpg_flags"]]
    return z, dqn, gdpg


def test_against_the_executed_reference_dqn_agent(engine, golden, all_models):
    """The drop-in DQNAgent against vectors the reference's OWN mwis_dqn_call.py produced (oracle/run_reference.py:
    the reference's Python executed with a NumPy stand-in for TensorFlow's ops): 8 shipped checkpoints x 4 graphs -
    act_values within 1e-5, the same argmax action, and solve_mwis (with and without zero weights: NetworkX pruning
    and id mapping in the reference) returning the same set, total and reward."""
    from distgcn_amd.mwis_dqn_call import DQNAgent
    z, dqn, _ = _ref_exec()
    worst = 0.0
    for m, fl in dqn:
        agent = DQNAgent(fl["feature_size"], flags=_flags(**fl))
        agent.model.set_params(all_models.params(m))
        for gi in (int(g) for g in z["graphs"]):
            adj, w = golden.scipy(gi), golden.csr(gi)[2]
            act_values, action = agent.predict(agent.makestate(adj, w.reshape(-1, 1)))
            ref = z["dqn|%s|g%02d|scores" % (m, gi)]
            worst = max(worst, float(np.abs(act_values - ref).max()))
            assert act_values.shape == ref.shape and np.abs(act_values - ref).max() <= 1e-5, (m, gi)
            assert action[0] == z["dqn|%s|g%02d|action" % (m, gi)][0] or \
                abs(ref[action[0], 0] - ref.max()) <= 2e-5, (m, gi)
            for tag in ("full", "zeros"):
                ww = z["dqn|%s|g%02d|%s|weights" % (m, gi, tag)]
                got, tot, reward = agent.solve_mwis(adj, ww)
                assert sorted(got) == z["dqn|%s|g%02d|%s|set" % (m, gi, tag)].tolist(), (m, gi, tag)
                assert tot == pytest.approx(float(z["dqn|%s|g%02d|%s|total" % (m, gi, tag)]), rel=1e-12) and reward == 1.0
    assert worst <= 1e-5


def test_against_the_executed_reference_at_full_size(engine, all_models):
    """tests/golden/ref_exec_big.npz (oracle/run_reference.py --big, the reference's own Python executed): (1) its
    mwis_dqn_call agent with the DQNBA l=20 checkpoint on graphs 1320 and 3945 of the C4 batch - BA N=300 m=2, where
    the float32 error tail of the whole 4 000-graph batch lives: act_values under conftest.check_scores' strict bar,
    the reference's own set and total from solve_mwis; (2) its mwis_gdpg_call agent with 20 layers (IS4SAT l=20 weights,
    GCN2_DQN: bias, activation on the last layer) on one ER G(500, 0.02) graph - the C5 shape: solve_mwis, solve_mwis_cit,
    solve_mwis_cgs_train (:778-839) and the b = 16 rollout (:596-659, its ties replayed from the seed) return the
    reference's sets and totals."""
    import json
    import scipy.sparse as sp
    from conftest import GOLDEN, check_scores
    from distgcn_amd import datagen
    from distgcn_amd.mwis_dqn_call import DQNAgent
    from distgcn_amd.mwis_gdpg_call import DQNAgent as GdpgAgent
    from oracle import ref_numpy as orc
    z = np.load(os.path.join(GOLDEN, "ref_exec_big.npz"))
    name = str(z["ba_model"])
    agent = DQNAgent(1, flags=_flags(num_layer=20))
    agent.model.set_params(all_models.params(name))
    layers = orc.gcn_layer_specs(all_models.params(name))
    for gi in (int(g) for g in z["ba_graphs"]):
        hb = datagen.ba_test2_batch(1, first_index=gi)
        adj = sp.csr_matrix((np.ones(hb.col_idx.size), hb.col_idx, hb.row_ptr), shape=(hb.num_nodes, hb.num_nodes))
        w = hb.weights
        act_values, action = agent.predict(agent.makestate(adj, w.reshape(-1, 1)))
        ref = z["ba|ba%04d|scores" % gi]
        f64, _ = orc.gcn_forward(layers, orc.makestate(adj, w.reshape(-1, 1), 1, 1, "dqn_call"), np.float64)
        check_scores(act_values, ref, f64, ("ba", gi), strict=True)
        got, tot, _ = agent.solve_mwis(adj, w)
        assert sorted(got) == z["ba|ba%04d|set" % gi].tolist(), gi
        assert tot == pytest.approx(float(z["ba|ba%04d|total" % gi]), rel=1e-12)
    fl = json.loads(str(z["c5_flags"]))
    g = GdpgAgent(_flags(**fl), seed=1)
    pre = "c5|var|"
    g.model.set_params({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    hb = datagen.er_batch(1, 500, 0.02)
    adj = sp.csr_matrix((np.ones(hb.col_idx.size), hb.col_idx, hb.row_ptr), shape=(500, 500))
    w = hb.weights
    vals, _ = g.predict(g.makestate(adj, w.reshape(-1, 1)))
    lay = orc.gcn_layer_specs({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}, model="GCN2_DQN", scope="model/gcn2_dqn")
    f64, _ = orc.gcn_forward(lay, orc.makestate(adj, w.reshape(-1, 1), 1, 1, "gdpg", "mwis"), np.float64)
    check_scores(vals, z["c5|scores"], f64, "c5", strict=True)
    for which in ("solve_mwis", "solve_mwis_cit", "solve_mwis_cgs_train"):
        got, tot = getattr(g, which)(adj, w)
        assert sorted(got) == z["c5|%s|set" % which].tolist(), which
        assert float(np.asarray(tot).ravel()[0]) == pytest.approx(float(z["c5|%s|total" % which]), rel=1e-12)
    np.random.seed(1234)
    got, tot = g.solve_mwis_rollout(adj, w, b=16, rng=np.random, reference_ties=True)
    assert sorted(got) == z["c5|solve_mwis_rollout|set"].tolist()
    assert float(np.asarray(tot).ravel()[0]) == pytest.approx(float(z["c5|solve_mwis_rollout|total"]), rel=1e-12)
    # the device-resident batched rollout (deterministic ties) is a maximal independent set at least as heavy as plain greedy
    dev, dtot = g.solve_mwis_rollout(adj, w, b=16)
    assert float(w[sorted(dev)].sum()) == pytest.approx(float(np.asarray(dtot).ravel()[0]), rel=1e-12)


def test_against_the_executed_reference_gdpg_solvers(engine, golden):
    """mwis_gdpg_call.DQNAgent (GCN2_DQN: bias on every layer, activation on the last) against the reference's own
    run: act_values within 1e-5; solve_mwis, solve_mwis_dit, solve_mwis_cit give the reference's sets and totals;
    the rollout family - where the reference draws np.random.choice among tied candidates - gives the reference's set
    whenever the deterministic tie rule (first candidate) reproduces it on the oracle, which must be the case for most;
    the per-component wrappers return the reference's TOTAL (its set mapping is off for one component of g01, see
    oracle/ref_numpy.solve_wrap)."""
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    z, _, gdpg = _ref_exec()
    agree = total = 0
    for ci, fl in enumerate(gdpg):
        agent = DQNAgent(_flags(**fl), seed=1)
        pre = "gdpg|%d|var|" % ci
        agent.model.set_params({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        layers = agent.model.layers
        fn = orc._default_scores_fn(layers, 1, 1, fl["predict"])
        for gi in (int(g) for g in z["gdpg_graphs"]):
            adj, w = golden.scipy(gi), golden.csr(gi)[2]
            vals, action = agent.predict(agent.makestate(adj, w.reshape(-1, 1)))
            assert np.abs(vals - z["gdpg|%d|g%02d|scores" % (ci, gi)]).max() <= 1e-5
            for name in ("solve_mwis", "solve_mwis_dit", "solve_mwis_cit", "solve_mwis_cgs_train"):
                got, tot = getattr(agent, name)(adj, w)
                assert sorted(got) == z["gdpg|%d|g%02d|%s|set" % (ci, gi, name)].tolist(), (ci, gi, name)
                assert float(np.asarray(tot).ravel()[0]) == pytest.approx(float(z["gdpg|%d|g%02d|%s|total" % (ci, gi, name)]), rel=1e-12)
            for name in ("solve_mwis_cit_wrap",):
                got, tot = getattr(agent, name)(adj, w)
                assert float(np.asarray(tot).ravel()[0]) == pytest.approx(float(z["gdpg|%d|g%02d|%s|total" % (ci, gi, name)]), rel=1e-12)
                assert float(w[sorted(got)].sum()) == pytest.approx(float(np.asarray(tot).ravel()[0]), rel=1e-12)
            variants = {"solve_mwis_rollout": dict(), "solve_mwis_rollout00": dict(rescore=False),
                        "solve_mwis_rollout0": dict(rescore=False, by_priority=True), "solve_mwis_rollout1": dict(by_priority=True)}
            for name, kw in variants.items():
                got, tot = getattr(agent, name)(adj, w, b=8)
                ref_set = z["gdpg|%d|g%02d|%s|set" % (ci, gi, name)].tolist()
                det, _ = orc.solve_mwis_rollout(fn, adj, w, b=8, predict=fl["predict"], **kw)  # first-candidate tie rule
                assert sorted(got) == sorted(det), (ci, gi, name)
                # and with the reference's own tie handling (exact ==, its summation order, np.random.choice
                # replayed from the seed oracle/run_reference.py used): the reference's set and total
                np.random.seed(1234)
                got_r, tot_r = getattr(agent, name)(adj, w, b=8, rng=np.random, reference_ties=True)
                total += 1
                agree += sorted(got_r) == ref_set and float(np.asarray(tot_r).ravel()[0]) == pytest.approx(
                    float(z["gdpg|%d|g%02d|%s|total" % (ci, gi, name)]), rel=1e-12)
    # (a float32 score ordering flip between the kernels and NumPy could legitimately change a candidate list)
    assert agree == total, (agree, total)


def test_c4_full_batch_and_its_eight_shards(engine, golden):
    """BASELINE config 4 at full size on ONE GPU: the 4 000-graph BA test2 batch with the trained DQNBA l=20 weights.
    Independence + maximality on all 4 000 sets; scores / sets bit-equal to the twin on every 10th graph; and the eight
    shards `parallel.shard_ranges` would hand to eight ranks, each solved as its own batch, reassemble to exactly the
    single-batch result (graphs are independent: sharding must not change a bit).  The device-resident sharded solve
    (`solve_sharded_device`, one rank here) returns the same."""
    from distgcn_amd import datagen, parallel
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = datagen.ba_test2_batch(4000)
    layers = golden.layers("result_DQNBA_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn")
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    res = engine.solve_fused(db, dm)
    engine.check_status(res["status"])
    state, totals, rounds = res["state"].cpu().numpy(), res["totals"].cpu().numpy(), res["rounds"].cpu().numpy()
    scores = res["scores"].cpu().numpy()
    import scipy.sparse as sp
    adj = sp.csr_matrix((np.ones(hb.col_idx.size, np.float32), hb.col_idx, hb.row_ptr), shape=(hb.num_nodes, hb.num_nodes))
    sel = (state == 1).astype(np.float32)
    hits = adj @ sel
    assert not np.any((hits > 0) & (state == 1)) and np.all((hits > 0) | (state == 1)) and not np.any(state == 0)
    sample = hb.select(range(0, 4000, 10))
    ref = ctwin.solve(sample, layers)
    k = 0
    for g in range(0, 4000, 10):
        n0, n1 = int(hb.graph_ptr[g]), int(hb.graph_ptr[g + 1])
        m = n1 - n0
        assert np.array_equal(state[n0:n1], ref["state"][k:k + m]) and rounds[g] == ref["rounds"][g // 10]
        assert np.array_equal(scores[n0:n1, 0].view(np.uint32), ref["scores"][k:k + m, 0].view(np.uint32))
        k += m
    ranges = parallel.shard_ranges(hb, 8)
    assert ranges[0][0] == 0 and ranges[-1][1] == 4000 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    cost = [int(hb.row_ptr[hb.graph_ptr[b]] - hb.row_ptr[hb.graph_ptr[a]] + hb.graph_ptr[b] - hb.graph_ptr[a]) for a, b in ranges]
    assert max(cost) <= 1.05 * (sum(cost) / 8)  # balanced on sum(nnz + N)
    for a, b in ranges:
        sub = hb.subset(a, b)
        r = engine.solve_fused(engine.upload(sub), dm)
        n0, n1 = int(hb.graph_ptr[a]), int(hb.graph_ptr[b])
        assert np.array_equal(r["state"].cpu().numpy(), state[n0:n1]) and np.array_equal(r["totals"].cpu().numpy(), totals[a:b])
    one = parallel.solve_sharded_device(engine, dm, hb)
    assert np.array_equal(one["state"], state) and np.array_equal(one["totals"], totals) and np.array_equal(one["rounds"], rounds)


@pytest.mark.parametrize("which", ["dit", "cit", "rollout", "rollout1"])
def test_iterative_solvers_cluster_variant_changes_nothing(engine, which, monkeypatch, cluster_switch):
    """The residual-graph kernel also runs as several workgroups per graph (small batches of deep stacks): same sets and
    totals as with one workgroup per graph, for graphs that fill several tiles, a 12-layer stack, weight features."""
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=12), seed=4)
    agent.device_iterative = True
    # (N = 500, the C5 size: 32 tiles - with K = 4 a workgroup owns eight, one per wave, and every wave aggregates two row sets)
    for hb, modes in ((datagen.er_batch(3, 200, 0.06, first_index=4400), ("0", "8", None)),
                      (datagen.er_batch(2, 500, 0.02, first_index=4410), ("0", "4", "5", None))):
        adjs = [hb.scipy_graph(g) for g in range(hb.num_graphs)]
        ws = [hb.weights[n0:n1] for n0, n1 in hb.graph_slices()]
        got = {}
        for mode in modes:
            cluster_switch(mode)
            got[mode] = agent.solve_iterative_batch(adjs, ws, which, b=8)
            assert got[mode] is not None
        for mode in modes[1:]:
            for a, b in zip(got["0"], got[mode]):
                assert a[0] == b[0] and np.array_equal(np.asarray(a[1]), np.asarray(b[1])), (which, mode, hb.num_nodes)


@pytest.mark.parametrize("which", ["cit", "rollout"])
def test_iterative_search_survives_a_cluster_fault(engine, which, cluster_switch):
    """Round 6: the residual launch takes the several-workgroups-per-graph form BY ITSELF on small batches (two 500-vertex graphs:
    K = 4, eight tiles per workgroup - C5's regime).  A placement fault in such a launch (injected: option test_cluster_fault)
    must not cost the search: the faulted step leaves every search it touched as it was (the state is written only when no
    workgroup of the launch has reported a fault), Engine.solve_residual switches the variant off, clears the bit and goes on -
    same final sets as with the variant off from the start, and the variant IS off afterwards."""
    from distgcn_amd import _lib, datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    lib = _lib.load()
    agent = DQNAgent(_flags(num_layer=12), seed=4)
    agent.device_iterative = True
    hb = datagen.er_batch(2, 500, 0.02, first_index=4410)
    adjs = [hb.scipy_graph(g) for g in range(hb.num_graphs)]
    ws = [hb.weights[n0:n1] for n0, n1 in hb.graph_slices()]
    cluster_switch(0)
    want = agent.solve_iterative_batch(adjs, ws, which, b=8)
    cluster_switch(None)  # automatic: the residual launch of this batch is a cluster launch
    auto = agent.solve_iterative_batch(adjs, ws, which, b=8)
    assert int(lib.dgcn_get_cluster()) == -1  # (no fault: still automatic)
    _lib.set_option("test_cluster_fault", 1)
    try:
        got = agent.solve_iterative_batch(adjs, ws, which, b=8)
    finally:
        _lib.set_option("test_cluster_fault", 0)
    assert int(lib.dgcn_get_cluster()) == 0  # the fault was seen (so the automatic choice WAS a cluster launch) and answered
    for a, b, c in zip(want, auto, got):
        assert a[0] == b[0] == c[0] and np.array_equal(np.asarray(a[1]), np.asarray(b[1])) and np.array_equal(np.asarray(a[1]), np.asarray(c[1])), which


def test_host_solver_compact_transfer(engine, monkeypatch):
    """Batches above the in-place threshold cross PCIe in the compact form (16-bit local column ids + degrees: include/dgcn.h
    DgcnCompactInfo) and are expanded on the device (csrc/expand.hip): same sets / rounds / totals / scores as the ordinary
    transfer (option host_compact = 0), as the expanded-on-the-device form (host_compact_direct = 0) and as the twin, on the BA mix (hubs, 100..300 vertices), an ER batch, and a batch with an
    unsorted row (entry order is the caller's, in both forms); the plain greedy search (no model) goes the same way."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.serving import HostSolver
    from oracle import ctwin
    layers = datagen.random_model(3, 32, seed=21)
    dm = DeviceModel(layers, engine.device)
    from distgcn_amd import _lib
    _lib.set_option("host_direct_bytes", 0)  # (conftest puts the table back after the test)

    def lists(hb, scramble=None):
        ps, cs, ws = [], [], []
        for g, (n0, n1) in enumerate(hb.graph_slices()):
            e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
            p, c = (hb.row_ptr[n0:n1 + 1] - e0).astype(np.int32), (hb.col_idx[e0:e1] - n0).astype(np.int32)
            if scramble == g:  # reverse the first row with at least two entries
                v = int(np.flatnonzero(np.diff(p) >= 2)[0])
                c = c.copy(); c[p[v]:p[v + 1]] = c[p[v]:p[v + 1]][::-1]
            ps.append(p); cs.append(c); ws.append(hb.weights[n0:n1].copy())
        return ps, cs, ws

    cases = [lists(datagen.ba_test2_batch(120, first_index=300)), lists(datagen.er_batch(150, 200, 0.1, first_index=7000)),
             lists(datagen.er_batch(150, 200, 0.1, first_index=7000), scramble=17)]
    for ps, cs, ws in cases:
        hb = HostBatch.from_csr_lists(ps, cs, ws)
        ref = ctwin.solve(hb, layers)
        # compact + read by the fused kernel as it is (the default), compact + expanded on the device first, ordinary transfer
        for mode, direct in (("1", "1"), ("1", "0"), ("0", "1")):
            _lib.set_option("host_compact", int(mode))
            _lib.set_option("host_compact_direct", int(direct))
            hs = HostSolver(engine, dm, depth=2, want_scores=True)
            g = [hs.solve(ps, cs, ws) for _ in range(2)][-1]
            hs.close()
            assert np.array_equal(g["state"], ref["state"]) and np.array_equal(g["rounds"], ref["rounds"]), (mode, direct)
            assert np.allclose(g["totals"], ref["totals"], rtol=1e-12, atol=0)
            assert np.array_equal(g["scores"].view(np.uint32), ref["scores"][:, 0].view(np.uint32)), (mode, direct)
        _lib.set_option("host_compact_direct", -1)
    ps, cs, ws = cases[0]
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    want = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, hb.weights, sum_weights=hb.weights, want_stats=False)
    _lib.set_option("host_compact", 1)
    hs = HostSolver(engine, None, depth=2)
    g = hs.solve(ps, cs, ws)
    hs.close()
    assert np.array_equal(g["state"], want["state"]) and np.array_equal(g["rounds"], want["rounds"])
