"""GPU: the any-size device path (csrc/general.hip) - what dgcn_solve_batch / dgcn_solve_residual_batch run for conflict
graphs beyond the fused kernel's 512 vertices / one CU's LDS (the multi-channel joint graphs of
wireless_dqn_test_mc.py:161, 244-289 have K * nflows vertices) and for layer stacks wider than 32.

Three kinds of checks: (1) forced onto shapes the fused kernel takes too (dgcn_set_general(1)), every step must equal
the fused kernel's bit for bit; (2) at N = 600 / 900 / 1 500 and on ER(500, 0.1) against the twin and the oracle's
solvers (oracle/ref_numpy.py, pinned by the executed reference) fed with the twin's scores; (3) the slot loop on a
K = 3 x 300-flow joint graph with all five schedulers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flags(**kw):
    from distgcn_amd.runtime_config import FLAGS
    base = dict(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1, predict="mwis")
    base.update(kw)
    return FLAGS.copy(**base)


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


@pytest.fixture
def general_switch():
    """dgcn_set_general for the duration of a test: 1 = every shape down the any-size path, None = automatic."""
    from distgcn_amd import _lib
    lib = _lib.load()
    initial = int(lib.dgcn_get_general())

    def set_to(value):
        lib.dgcn_set_general(-1 if value is None else int(value))
    yield set_to
    lib.dgcn_set_general(initial)


def _scipy(hb, g):
    m = hb.scipy_graph(g).tocsr()
    m.sort_indices()
    return m


@pytest.mark.parametrize("shape", [(3, 32, False), (1, 32, False), (4, 16, True), (2, 48, False), (20, 32, False), (3, 64, True)])
def test_plain_solve_equals_fused_and_twin(engine, golden, general_switch, shape):
    """dgcn_solve_batch down the any-size path on fixture graphs: scores, sets, rounds, totals bit-equal to the fused
    kernel's (where it takes the model) and to the twin's."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    num_layer, hidden, bias = shape
    layers = datagen.random_model(num_layer, hidden, bias=bias, seed=11 + num_layer)
    hb = golden.host_batch([0, 3, 9, 12, 5, 1])
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    ref = ctwin.solve(hb, layers)
    general_switch(None)
    fused = None
    if engine.solve_path(db, dm) == 1:
        r = engine.solve_fused(db, dm)
        engine.check_status(r["status"])
        fused = {k: r[k].cpu().numpy() for k in ("state", "scores", "rounds", "totals")}
    general_switch(1)
    assert engine.solve_path(db, dm) == 2
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    got = {k: r[k].cpu().numpy() for k in ("state", "scores", "rounds", "totals")}
    assert np.array_equal(got["scores"].ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(got["state"], ref["state"])
    assert np.array_equal(got["rounds"], ref["rounds"])
    assert np.allclose(got["totals"], ref["totals"], rtol=1e-12)
    if fused is not None:
        assert np.array_equal(got["scores"].ravel().view(np.uint32), fused["scores"].ravel().view(np.uint32))
        assert np.array_equal(got["state"], fused["state"]) and np.array_equal(got["rounds"], fused["rounds"])
        assert np.allclose(got["totals"], fused["totals"], rtol=1e-12)


STEPPERS = {  # which -> (greedy mode name, max_rounds, options: scores given / completions by priority, predict)
    "dit": ("GREEDY_ROUNDS", 1, (False, False), "mwis"),
    "lgs_all": ("GREEDY_ROUNDS", 0, (False, False), "mwis"),
    "cit": ("GREEDY_CENTRAL", 1, (False, False), "mwis"),
    "rollout": ("GREEDY_ROLLOUT", 1, (False, False), "mwis"),
    "rollout00": ("GREEDY_ROLLOUT", 1, (True, False), "mwis"),
    "rollout0": ("GREEDY_ROLLOUT", 1, (True, True), "mwis"),
    "rollout1": ("GREEDY_ROLLOUT", 1, (False, True), "mwis"),
    "dit_mis": ("GREEDY_ROUNDS", 1, (False, False), "mis"),
    "rollout_mis": ("GREEDY_ROLLOUT", 1, (False, False), "mis"),
}


@pytest.mark.parametrize("which", sorted(STEPPERS))
def test_residual_steps_equal_the_fused_kernel(engine, golden, general_switch, which):
    """One call of dgcn_solve_residual_batch = one solver step of the whole batch.  Step by step, from a start with
    decided vertices, a graph with nothing left and a graph without positive weight, the any-size path must leave the
    same state bytes, scores (bits), rounds and totals as the fused residual-graph kernel, and stop after as many steps."""
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    mode_name, max_rounds, (given, by_prio), predict = STEPPERS[which]
    agent = DQNAgent(_flags(num_layer=5, predict=predict), seed=9)  # (three hidden aggregations: the one-launch stack of big.hip)
    rng = np.random.default_rng(5)
    for k in agent.model.vars:  # non-zero biases
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    hb = golden.host_batch([2, 7, 1, 0, 12, 8])
    sl = hb.graph_slices()
    hb.weights[sl[4][0]:sl[4][1]] = 0.0            # a graph without positive weight: left alone
    hb.weights[sl[2][0]:sl[2][0] + 5] = 0.0        # some zero weights inside a live graph
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    init = np.where(rng.random(hb.num_nodes) < 0.2, rng.integers(1, 3, hb.num_nodes), 0).astype(np.uint8)
    init[sl[3][0]:sl[3][1]] = 2                    # a graph with nothing left
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
        if not a[1].any():  # no graph ran a round: the search is over
            break
        assert steps < 400
    assert steps > 1
    st = snap[1][0]
    assert np.array_equal(st[sl[4][0]:sl[4][1]], init[sl[4][0]:sl[4][1]])  # the weightless graph was left alone
    for path in (None, 1):  # and the host loop (progress words) stops after the same number of steps on both paths
        general_switch(path)
        s0 = torch.from_numpy(init.copy()).to(engine.device)
        res = engine.solve_residual(db, dm, s0, predict=predict, greedy=greedy, max_rounds=max_rounds, beam=6,
                                    weight_features=predict != "mwis", options=options, finish_small=False,
                                    scores=None if full_scores is None else full_scores.clone())
        assert res["steps"] == steps - 1 and np.array_equal(s0.cpu().numpy(), st), (which, path, res["steps"], steps)


BIG = [(600, 0.01), (500, 0.1), (900, 0.02), (1500, 0.004), (700, 0.15), (976, 0.05)]


@pytest.mark.parametrize("n,p", BIG)
def test_big_graphs_plain_solve_vs_twin(engine, n, p):
    """Graphs the fused kernel cannot hold: ER(500, 0.1) (25 000 entries), 600, 900 and 1 500 vertices, dense graphs whose
    column lists alone exceed the LDS (ER(700, 0.15): 73 000 entries; k_big's largest, 976 vertices at 47 000) - one call of
    dgcn_solve_batch, scores / sets / rounds equal to the twin bit for bit, through the agent API and the host solver."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.mwis_dqn_call import DQNAgent, solve_csr_lists
    from oracle import ctwin
    hb = datagen.er_batch(3, n, p, first_index=40)
    agent = DQNAgent(1, flags=_flags(num_layer=4))
    dm = DeviceModel(agent.model.layers, engine.device)
    db = engine.upload(hb)
    assert not engine.solve_supported(db, dm) and engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, agent.model.layers)
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"])
    assert np.allclose(r["totals"].cpu().numpy(), ref["totals"], rtol=1e-12)
    # the reference's call pattern: per-graph CSR arrays in, sets out (HostSolver -> dgcn_solve_batch)
    ps = [hb.row_ptr[hb.graph_ptr[g]:hb.graph_ptr[g + 1] + 1].astype(np.int64) - int(hb.row_ptr[hb.graph_ptr[g]]) for g in range(3)]
    cs = [hb.col_idx[hb.row_ptr[hb.graph_ptr[g]]:hb.row_ptr[hb.graph_ptr[g + 1]]].astype(np.int64) - int(hb.graph_ptr[g]) for g in range(3)]
    ws = [hb.weights[hb.graph_ptr[g]:hb.graph_ptr[g + 1]] for g in range(3)]
    res, gp = solve_csr_lists(engine, agent.model, ps, cs, ws, "mwis", "auto")
    assert np.array_equal(res["state"], ref["state"])
    assert np.array_equal(np.asarray(res["scores"]).ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))


def _hub_graph(n, hubs, p, rng):
    """ER(n, p) plus `hubs` = [(vertex, degree), ..]: rows far longer than every other row of their sixteen-row tile."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    indptr, indices = datagen.er_graph(n, p, rng)
    a = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n)).tolil()
    for v, d in hubs:
        for u in rng.choice(np.setdiff1d(np.arange(n), [v]), size=d, replace=False):
            a[v, u] = 1.0
            a[u, v] = 1.0
    a = a.tocsr()
    a.sort_indices()
    return a


@pytest.mark.parametrize("case", ["er900_p0.7", "two_hubs", "hub_classes"])
def test_big_rows_beyond_575_entries_vs_twin(engine, case):
    """Rows of 575 and more entries (round 4's row-order counting sort clamped entry counts to 576 bins and sized a tile's
    trips from its FIRST row: a longer row later in the tile lost the tail of its sum in every layer - advisor finding,
    round 4).  ER(900, 0.7): every row has ~630 entries; two hubs of degree 600 and 900 in a sparse 950-vertex graph;
    hubs of 576 .. 940 entries sharing tiles.  Scores, sets and rounds equal to the twin bit for bit, no fault bit."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(77)
    if case == "er900_p0.7":
        hb = datagen.er_batch(2, 900, 0.7, first_index=90)
    else:
        if case == "two_hubs":
            mats = [_hub_graph(950, [(17, 600), (640, 900)], 0.01, rng), _hub_graph(930, [(3, 929), (4, 580)], 0.004, rng)]
        else:
            mats = [_hub_graph(976, [(5 + 31 * i, 576 + 19 * i) for i in range(20)], 0.01, rng)]
        hb = HostBatch.from_scipy(mats, [rng.random(m.shape[0]) for m in mats])
    layers = datagen.random_model(4, 32, seed=5)
    dm = DeviceModel(layers, engine.device)
    db = engine.upload(hb)
    assert engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, layers)
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"])


@pytest.mark.parametrize("which", ["dit", "cit", "rollout", "rollout1"])
def test_big_graph_iterative_solvers_vs_oracle(engine, which):
    """solve_mwis_dit / _cit / _rollout on a 600-vertex graph and on ER(500, 0.1): entirely on the device (the any-size
    path), decisions equal to the oracle's solvers fed with the TWIN's scores (`_twin_scores_fn`): the control flow is what
    this pins - the forward is two hops from the reference here (HIP == twin bit for bit; twin vs restatement: the parity
    tests); test_big_graph_iterative_solvers_restatement_forward runs the restatement's own forward."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=3), seed=21)
    fn = _twin_scores_fn(agent.model.layers)
    cases = [(600, 0.01, 3)] + ([(500, 0.1, 4)] if which != "cit" else [])
    for n, p, seed in cases:
        rng = np.random.default_rng(20230800 + seed)
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


@pytest.mark.parametrize("which", ["dit", "cit", "rollout"])
def test_big_graph_iterative_solvers_restatement_forward(engine, which):
    """test_big_graph_iterative_solvers_vs_oracle feeds the oracle's solvers with the TWIN's scores (control flow pinned by
    the executed reference, the forward two hops away).  Here the same 600-vertex search against the oracle's solvers fed
    with the restatement's own float32 forward (oracle/ref_numpy._default_scores_fn - the NumPy restatement of
    gcn/layers.py:189-216, mwis_gdpg_call.py:211-216): one hop.  (On this graph no decision lies inside the two forwards'
    rounding distance - the oracle's solvers select the same sets with either forward, checked on the CPU.)"""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=3), seed=21)
    fn = orc._default_scores_fn(agent.model.layers)
    n, p = 600, 0.01
    rng = np.random.default_rng(20230800 + 3)
    indptr, indices = datagen.er_graph(n, p, rng)
    adj = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    w = rng.random(n)
    got = agent.solve_iterative_batch([adj], [w], which, b=4)
    assert got is not None
    want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else \
        orc.solve_mwis_rollout(fn, adj, w, b=4)
    assert got[0][0] == want[0], which
    assert np.allclose(got[0][1], want[1], rtol=1e-12)


@pytest.mark.parametrize("n,p", [(900, 0.01), (1500, 0.004)])
def test_big_graph_iterative_solvers_vs_host_reslicing(engine, n, p):
    """N = 900 and 1 500, three graphs advanced together: the device-resident solvers against the reference's own
    control flow (SciPy re-slicing per step on the host, mwis_gdpg_call.py:278-318, 343-384, 596-659, with every forward
    pass and completion on the device) - same sets, same totals."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=3), seed=2)
    adjs, wts = [], []
    for g in range(3):
        rng = np.random.default_rng(20230900 + g)
        indptr, indices = datagen.er_graph(n - 7 * g, p, rng)  # ragged
        adjs.append(sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(n - 7 * g,) * 2))
        wts.append(rng.random(n - 7 * g))
    for which, host in (("dit", "solve_mwis_dit"), ("cit", "solve_mwis_cit"), ("rollout", "solve_mwis_rollout")):
        agent.device_iterative = True
        got = agent.solve_iterative_batch(adjs, wts, which, b=3)
        assert got is not None and len(got) == 3
        agent.device_iterative = False
        for g in ((0, 2) if which == "dit" else (1,)):
            want = getattr(agent, host)(adjs[g], wts[g], b=3) if which == "rollout" else getattr(agent, host)(adjs[g], wts[g])
            assert got[g][0] == want[0], (which, g)
            assert np.allclose(got[g][1], want[1], rtol=1e-12)
        for (sel, _), adj in zip(got, adjs):  # independent and maximal
            m = np.zeros(adj.shape[0], bool); m[list(sel)] = True
            assert not (adj[m][:, m]).nnz and np.all((adj @ m.astype(np.float64) > 0) | m)


@pytest.mark.parametrize("algo", ["Greedy", "DGCN-LGS", "DGCN-LGS-it", "CGCN-CGS", "DGCN-RS"])
def test_wireless_joint_multichannel_graph_900(engine, algo):
    """The multi-channel experiment's shape (bash/twc_major_wireless_mc_test.sh: num_channels = 3): 300 flows, the joint
    conflict graph on 3 x 300 vertices built by wireless.multichannel_conflict_graph (wireless_rollout_test_flood.py:98-133),
    two instances in lockstep, every scheduler of wireless_dqn_test_mc.py:236-289 - against the per-instance restatement of
    the slot loop with the oracle's solvers (DGCN-RS: the agent's host re-slicing control flow) as schedulers."""
    import scipy.sparse as sp
    from distgcn_amd import datagen, wireless
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_dqn_call import DQNAgent as DqnAgent
    from distgcn_amd.mwis_gdpg_call import DQNAgent as GdpgAgent
    from oracle import ctwin, ref_numpy as orc, ref_wireless
    nflows, K, T = 300, 3, 4
    adjs, traffics = [], []
    for i in range(2):
        rng = np.random.default_rng(500 + i)
        indptr, indices = datagen.er_graph(nflows, 0.02, rng)
        base = sp.csr_matrix((np.ones(indices.size), indices, indptr), shape=(nflows, nflows))
        chans = wireless.multichannel_conflict_simulate(base, k=K, p=0.8, rng=np.random.RandomState(7 + i))
        _, joint = wireless.multichannel_conflict_graph(chans)
        assert joint.shape == (K * nflows, K * nflows)
        adjs.append(joint)
        traffics.append(wireless.make_traffic(nflows, T, 0.05, n_ch=K, seed=30 + i))
    if algo in ("Greedy", "DGCN-LGS"):
        agent = DqnAgent(1, flags=_flags(num_layer=3))
    else:
        agent = GdpgAgent(_flags(num_layer=3), seed=4)
    layers = agent.model.layers
    fn = _twin_scores_fn(layers)

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

    def rs_fn(adj, w):
        if not w.size:
            return set()
        agent.device_iterative = False
        try:
            return agent.solve_mwis_rollout_wrap(adj, w, b=16)[0]
        finally:
            agent.device_iterative = True

    solver = {"Greedy": greedy_fn, "DGCN-LGS": dgcn_fn,
              "DGCN-LGS-it": lambda a, w: orc.solve_mwis_dit(fn, a, w)[0] if w.size else set(),
              "CGCN-CGS": lambda a, w: orc.solve_mwis_cgs_train(fn, a, w)[0] if w.size else set(),
              "DGCN-RS": rs_fn}[algo]
    got = wireless.simulate(adjs, traffics, algo=algo, agent=agent, wt_sel="qr")
    for i in range(len(adjs)):
        want = ref_wireless.simulate_one(adjs[i], traffics[i]["arrival_pkts"], traffics[i]["link_rates"], solver, "qr")
        assert np.array_equal(got[i]["queue"], want["queue"]), (algo, i)
        assert np.array_equal(got[i]["depart"], want["depart"]), (algo, i)
        assert np.allclose(got[i]["total_wt"], want["total_wt"], rtol=1e-12)
        assert got[i]["depart"].sum() > 0


def test_wide_deep_model_runs_on_the_device(engine, golden):
    """A deep stack with hidden width 64 (outside k_fused's 32): dgcn_solve_batch and the residual solvers take the
    any-size path instead of returning to the host; same bits as the twin / the oracle."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ctwin, ref_numpy as orc
    layers = datagen.random_model(5, 64, seed=3)
    hb = golden.host_batch([2, 7, 1])
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    assert engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, layers)
    r = engine.solve_fused(db, dm)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), ref["state"])
    agent = DQNAgent(_flags(num_layer=4, hidden1=48), seed=3)
    fn = _twin_scores_fn(agent.model.layers)
    adj, w = golden.scipy(2), golden.csr(2)[2]
    got = agent.solve_iterative_batch([adj], [w], "cit")
    assert got is not None
    want = orc.solve_mwis_cit(fn, adj, w)
    assert got[0][0] == want[0]


def test_residual_step_in_one_launch_agrees_with_the_compaction_path(engine, tmp_path):
    """A residual step of a deep model on graphs k_big takes is ONE launch (big.hip: the residual graph's support from the
    adjacency and the running state, every layer, the greedy step).  Second witness besides the fused kernel (the step-by-step
    test above, on fixture graphs): complete dit / cit / rollout searches on three ragged ~900-vertex graphs - zero weights
    inside live graphs, biases, a leaky last layer - by a child process as built and by one with option big_residual = 0, the
    compaction launches + k_big + k_lgs that ran before: same states, step counts, score bits."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wide_witness.py")
    files = {}
    for tag, val in (("one_launch", "1"), ("compaction", "0")):
        files[tag] = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, script, files[tag], "5"], check=True, env=dict(os.environ, DGCN_OPTIONS="big_residual=" + val), timeout=900)
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


@pytest.mark.parametrize("num_layer", [1, 5])
@pytest.mark.parametrize("beam", [1, 3, 64])
def test_rollout_beam_edge_values_equal_the_fused_kernel(engine, golden, general_switch, num_layer, beam):
    """The rollout's candidates are selected at the end of the step's own launch on the any-size path (cand_select.h, called by
    k_big / k_wide1) and the instances' masks are made by k_lgs: with one candidate, with as many as the list holds (64) and
    late in a search - fewer undecided vertices than candidates - the searches end in the same states after the same number
    of steps as the fused kernel's.  (More than 64 candidates: refused by the entry point on both paths.)"""
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=num_layer), seed=4)
    hb = golden.host_batch([2, 7, 1, 0, 12, 8])
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    from distgcn_amd._lib import DgcnError
    results = {}
    for path in (None, 1):
        general_switch(path)
        assert engine.solve_path(db, dm) == (1 if path is None else 2)
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
        with pytest.raises(DgcnError, match="beam"):
            engine.solve_residual(db, dm, state, greedy=engine.GREEDY_ROLLOUT, max_rounds=1, beam=65, max_steps=1)
        res = engine.solve_residual(db, dm, state, greedy=engine.GREEDY_ROLLOUT, max_rounds=1, beam=beam, finish_small=False)
        engine.check_status(res["status"])
        results[path] = (state.cpu().numpy().copy(), res["steps"])
    assert results[None][1] == results[1][1] and results[1][1] > 3
    assert np.array_equal(results[None][0], results[1][0])
    assert not (results[1][0] == 0).any()


@pytest.mark.parametrize("num_layer", [1, 5])
def test_rollout_nan_weight_on_the_any_size_path(engine, golden, general_switch, num_layer):
    """A NaN weight makes a NaN priority: the reference would spin on it; here the graph is reported (DGCN_FAULT_NAN_PRIORITY),
    gets no candidates and keeps its state, and the other graphs of the batch finish their searches as without it."""
    import torch
    from distgcn_amd import _lib
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=num_layer), seed=4)
    hb = golden.host_batch([2, 7, 1])
    sl = hb.graph_slices()
    clean = engine.upload(hb)
    dm = agent.model.device_model(engine)
    general_switch(1)
    hb.weights[sl[1][0] + 3] = np.nan
    db = engine.upload(hb)
    s0 = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
    s1 = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
    out0, out1 = engine.solve_buffers(clean, False), engine.solve_buffers(db, False)
    for _ in range(70):  # (single steps: the host loop of a whole search stops at the first fault it reads back)
        engine.solve_residual(clean, dm, s0, greedy=engine.GREEDY_ROLLOUT, max_rounds=1, beam=5, max_steps=1, out=out0)
        engine.solve_residual(db, dm, s1, greedy=engine.GREEDY_ROLLOUT, max_rounds=1, beam=5, max_steps=1, out=out1)
    engine.check_status(out0["status"])
    assert int(out1["status"].cpu().numpy().ravel()[0]) & _lib.FAULT_NAN_PRIORITY
    a, b = s0.cpu().numpy(), s1.cpu().numpy()
    assert not (a == 0).any()
    assert not b[sl[1][0]:sl[1][1]].any()                                 # the graph with the NaN: untouched
    for g in (0, 2):
        assert np.array_equal(a[sl[g][0]:sl[g][1]], b[sl[g][0]:sl[g][1]])  # the others: as without it


@pytest.mark.parametrize("num_layer,nodes", [(1, 900), (2, 900), (5, 900), (5, 1500)])
def test_rollout_in_one_launch_agrees_with_the_instance_launches(engine, tmp_path, num_layer, nodes):
    """Witness for csrc/rollout_bits.h: complete searches on three ragged graphs (zero weights inside live graphs) by a child
    process as built - candidates, all completions (an instance per bit) and the pick inside the step's launch of k_wide1 /
    k_big / k_big2 - and by one with option rollout_bits = 0 - k_lgs on beam x graphs masked instances + k_res_pick: the same
    states after the same number of steps (and every other solver's results untouched by the switch)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wide_witness.py")
    files = {}
    for tag, val in (("bits", "1"), ("launches", "0")):
        files[tag] = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, DGCN_OPTIONS="rollout_bits=" + val)
        subprocess.run([sys.executable, script, files[tag], str(num_layer), str(nodes)], check=True, env=env, timeout=600)
    a, b = np.load(files["bits"]), np.load(files["launches"])
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith("_scores"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        elif k.endswith("_totals"):
            assert np.allclose(a[k], b[k], rtol=1e-12), k
        else:
            assert np.array_equal(a[k], b[k]), k
    assert a["rollout_steps"][0] > 50 and not (a["rollout_state"] == 0).any()


@pytest.mark.parametrize("num_layer,nodes", [(1, 900), (2, 900), (5, 1500)])
def test_rounds_on_ahead_lists_agree_with_the_three_phase_rounds(engine, tmp_path, num_layer, nodes):
    """Witness for lgs_rounds_ahead (lgs_rounds.h): the plain solve (states, round counts, totals, score bits) and complete
    searches of k_wide1 / k_big2 by a child process as built - a whole search's rounds as two walks over every vertex's AHEAD
    list - and by one with option wide_ahead = 0 - lgs_rounds' three phases over all neighbours: the same bytes."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wide_witness.py")
    files = {}
    for tag, val in (("ahead", "1"), ("rounds", "0")):
        files[tag] = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, DGCN_OPTIONS="wide_ahead=" + val)
        subprocess.run([sys.executable, script, files[tag], str(num_layer), str(nodes)], check=True, env=env, timeout=600)
    a, b = np.load(files["ahead"]), np.load(files["rounds"])
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith("_scores"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        elif k.endswith("_totals"):
            assert np.allclose(a[k], b[k], rtol=1e-12), k
        else:
            assert np.array_equal(a[k], b[k]), k
    assert a["plain_rounds"].min() >= 2


@pytest.mark.parametrize("hidden,num_layer,n,p", [(16, 20, 600, 0.012), (16, 4, 900, 0.01), (4, 4, 900, 0.01), (3, 3, 700, 0.02), (2, 3, 520, 0.05),
                                                  (16, 5, 1500, 0.006), (5, 3, 1200, 0.01)])
def test_narrow_deep_stacks_take_the_one_launch_kernels(engine, hidden, num_layer, n, p):
    """Round-5 review item 4a.  The reference ships deep stacks narrower than 32 (model/result_IS4SAT_deep_ld1_c16_l20.., c16_l4,
    c4_l4, c3_l3, c2_l3: widths from --hidden1, gcn/models.py:550-573) and runs them on any conflict graph it gets
    (wireless_dqn_test_mc.py:161).  Beyond the fused kernel's 512 vertices they ran layer by layer (a 20-launch chain); now as
    their 32-wide zero-padded copy on k_big / k_big2: ONE launch for the plain solve, bit-equal to the twin (which computes at
    the model's own width) - scores, sets, rounds, totals - on 520 .. 1 500 vertices, biases and a leaky last layer included."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    layers = datagen.random_model(num_layer, hidden, bias=True, last_act="leaky_relu", seed=31 + hidden)
    hb = datagen.er_batch(3, n, p, first_index=70 + hidden)
    sl = hb.graph_slices()
    hb.weights[sl[1][0]:sl[1][0] + 7] = 0.0
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    assert engine.solve_path(db, dm) == 2
    ref = ctwin.solve(hb, layers)
    from distgcn_amd import _lib
    # (k_big2's lone workgroup per CU pays from a device-filling batch on - csrc/fused.hip narrow_wants_pad: with three graphs
    # the automatic choice beyond 976 vertices is the chain; option narrow_pad = 1 takes the copy wherever a kernel takes it)
    for forced in ((-1,) if n <= 976 else (-1, 1)):
        with _lib.options(narrow_pad=forced):
            engine.timing(True)
            r = engine.solve_fused(db, dm)
            engine.torch.cuda.synchronize()
            engine.timing(False)
        engine.check_status(r["status"])
        assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), ref["scores"].ravel().view(np.uint32)), forced
        assert np.array_equal(r["state"].cpu().numpy(), ref["state"]), forced
        assert np.array_equal(r["rounds"].cpu().numpy(), ref["rounds"]), forced
        assert np.allclose(r["totals"].cpu().numpy(), ref["totals"], rtol=1e-12)
        if n <= 976 or forced == 1:
            assert engine.timing_read("big_solve")[1] == 1  # the whole path in one launch of k_big / k_big2 ...
            assert engine.timing_read("fused_pad")[1] == 1  # ... behind the one that writes the padded copy
            assert engine.timing_read("spmm")[1] == 0 and engine.timing_read("transform")[1] == 0  # ... and no layer-by-layer kernel
        else:
            assert engine.timing_read("big_solve")[1] == 0 and engine.timing_read("spmm")[1] > 0  # three graphs: the chain
    # one hop from the reference: the restatement's own float32 / float64 forward of the NARROW model (oracle/ref_numpy.py: the
    # NumPy restatement of gcn/layers.py:189-216) on the smallest graph - the bar of conftest.check_scores, and the same set
    # wherever the margin test proves it (no excluded vertex within twice the measured error of a member neighbour)
    from conftest import check_scores
    from oracle import ref_numpy as orc
    g = int(np.argmin([b - a for a, b in sl]))
    n0, n1 = sl[g]
    adj, w = _scipy(hb, g), hb.weights[n0:n1]
    state = orc.makestate(adj, w.reshape(-1, 1), 1, 1, "gdpg", "mwis")
    f32 = orc.gcn_forward(layers, state, np.float32)[0][:, 0]
    f64 = orc.gcn_forward(layers, state, np.float64)[0][:, 0]
    got = r["scores"].cpu().numpy().ravel()[n0:n1]
    e32, e64, _ = check_scores(got, f32, f64, what=("narrow", hidden, num_layer, n))
    want_set, _ = orc.local_greedy_search(adj, f32.astype(np.float64) * w)
    mine = set(np.flatnonzero(r["state"].cpu().numpy()[n0:n1] == 1).tolist())
    if mine != set(int(v) for v in want_set):  # admissible only as a flagged near-tie
        pr = got.astype(np.float64) * w
        delta = 2 * max(e32, 1e-7)
        st = r["state"].cpu().numpy()[n0:n1]
        risky = 0
        for v in np.flatnonzero(st == 2):
            nb = adj.indices[adj.indptr[v]:adj.indptr[v + 1]]
            members = nb[st[nb] == 1]
            if not any(pr[u] - pr[v] > delta * (abs(w[u]) + abs(w[v])) for u in members):
                risky += 1
        assert risky > 0, ("sets differ without a margin-flagged vertex", hidden, num_layer, n)


@pytest.mark.parametrize("hidden,num_layer,n", [(16, 4, 900), (4, 4, 600), (16, 5, 1200)])
def test_narrow_deep_stacks_iterative_solvers_agree_with_the_layer_by_layer_path(engine, hidden, num_layer, n):
    """... and their residual steps: complete dit / cit / rollout searches (b = 16) on three ragged graphs as built - padded,
    every step one launch of k_big<RESID> / k_big2<RESID>, the tail of the search in k_tail - and with k_big / k_big2 switched
    off (options big = 0, big2 = 0: the model's own width through the compaction launches and the layer-by-layer kernels, what
    ran before): same final states and totals; and against the oracle's solvers fed with the twin's scores."""
    import torch
    from distgcn_amd import datagen, _lib
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    import scipy.sparse as sp
    from oracle import ref_numpy as orc
    layers = datagen.random_model(num_layer, hidden, bias=True, last_act="leaky_relu", seed=5 + hidden)
    rng = np.random.default_rng(123 + n)
    mats, ws = [], []
    for nn in (n, n - 37, n - 210):
        ip, ix = datagen.er_graph(nn, 9.0 / nn, rng)
        mats.append(sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(nn, nn)))
        w = rng.random(nn)
        w[rng.random(nn) < 0.04] = 0.0
        ws.append(w)
    hb = HostBatch.from_scipy(mats, ws)
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    assert engine.solve_path(db, dm) == 2
    fn = _twin_scores_fn(layers)
    for name, kw in (("dit", dict(greedy=engine.GREEDY_ROUNDS, max_rounds=1)), ("cit", dict(greedy=engine.GREEDY_CENTRAL)),
                     ("rollout", dict(greedy=engine.GREEDY_ROLLOUT, beam=16))):
        got = {}
        for tag, opts in (("padded", {}), ("layered", dict(big=0, big2=0))):
            with _lib.options(**opts):
                state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
                res = engine.solve_residual(db, dm, state, **kw)
                engine.check_status(res["status"])
                got[tag] = state.cpu().numpy().copy()
        assert np.array_equal(got["padded"], got["layered"]), name
        assert (got["padded"] == 1).sum() > 50  # (a search happened; zero-weight vertices may stay undecided: np.sum(wts_nn) <= 0 -> break)
        g = 2  # the smallest graph against the oracle's solver (pure Python: seconds)
        n0, n1 = hb.graph_slices()[g]
        solver = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit, "rollout": orc.solve_mwis_rollout}[name]
        want, _ = solver(fn, mats[g], ws[g], 16) if name == "rollout" else solver(fn, mats[g], ws[g])
        assert set(np.flatnonzero(got["padded"][n0:n1] == 1).tolist()) == set(int(v) for v in want), name


def _twin_scores_fn_poly(layers):
    """scores_fn for the oracle's solvers on an [I, L, L.L] model: the twin's forward on the residual graph's own L and L.L."""
    from distgcn_amd.batch import HostBatch
    from oracle import ctwin
    import scipy.sparse as sp

    def fn(adj_nn, wts_nn):
        a = sp.csr_matrix(adj_nn)
        a.sort_indices()
        hb = HostBatch.from_csr_lists([a.indptr.astype(np.int64)], [a.indices.astype(np.int64)])
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        lap2 = ctwin.supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        return ctwin.forward([lap, lap2], layers, hb.num_nodes)
    return fn


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (3, 8), (4, 32)])
def test_three_support_models_in_the_solve_entry_points(engine, golden, all_models, shape):
    """Round-5 review item 4b.  [I, L, L.L] models (max_degree = 2: gcn/utils.py:268-271, gcn/layers.py:199-208; the two shipped
    ..cheb2.. checkpoints are c1_l1 and c1_l2) go through dgcn_solve_batch / dgcn_solve_residual_batch like every other model:
    L.L formed on the device into a workspace slice bounded on the host (no read-back between the count and the fill pass),
    the layer-by-layer forward, the greedy step - ONE C call, nothing composed in Python.  Plain solve: scores bit-equal to
    the twin's forward on the twin's L and L.L, sets / rounds / totals equal to the twin's search; shipped weights for the
    two shipped shapes, random ones for deeper / wider stacks; ragged fixture graphs with an isolated vertex and zero weights.
    Iterative solvers (dit / cit / rollout): final sets equal to the oracle's solvers fed with the twin's scores."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    from oracle import ref_numpy as orc
    num_layer, hidden = shape
    shipped = {(1, 1): "result_IS4SAT_deep_ld1_c1_l1_cheb2_diver1_mwis_dqn", (2, 1): "result_IS4SAT_deep_ld1_c1_l2_cheb2_diver1_mwis_dqn"}
    layers = all_models.layers(shipped[shape]) if shape in shipped else datagen.random_model(num_layer, hidden, bias=True, seed=77, num_supports=3)
    assert len(layers[0]["weights"]) == 3
    ids = [0, 3, 9, 12, 5]
    hb = golden.host_batch(ids)
    sl = hb.graph_slices()
    hb.weights[sl[2][0]:sl[2][0] + 4] = 0.0
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    assert not engine.solve_supported(db, dm) and engine.solve_path(db, dm) == 2
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    lap2 = ctwin.supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    want_sc = ctwin.forward([lap, lap2], layers, hb.num_nodes)
    prio = want_sc[:, 0].astype(np.float64) * hb.weights
    want = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, prio, sum_weights=hb.weights, want_stats=False)
    engine.timing(True)
    r = engine.solve_fused(db, dm)
    engine.torch.cuda.synchronize()
    engine.timing(False)
    engine.check_status(r["status"])
    assert np.array_equal(r["scores"].cpu().numpy().ravel().view(np.uint32), want_sc.ravel().view(np.uint32))
    assert np.array_equal(r["state"].cpu().numpy(), want["state"]) and np.array_equal(r["rounds"].cpu().numpy(), want["rounds"])
    assert np.allclose(r["totals"].cpu().numpy(), want["totals"], rtol=1e-12)
    assert engine.timing_read("supports2_count")[1] == 1 and engine.timing_read("supports2_fill")[1] == 1
    # the composition in Python that this replaces (forward through the poly entry + dgcn_lgs_batch): same bits
    old = engine.solve(db, dm, mode=0)
    assert np.array_equal(old["scores"].cpu().numpy().ravel().view(np.uint32), r["scores"].cpu().numpy().ravel().view(np.uint32))
    assert np.array_equal(old["state"].cpu().numpy(), r["state"].cpu().numpy())
    fn = _twin_scores_fn_poly(layers)
    for name, kw in (("dit", dict(greedy=engine.GREEDY_ROUNDS, max_rounds=1)), ("cit", dict(greedy=engine.GREEDY_CENTRAL)),
                     ("rollout", dict(greedy=engine.GREEDY_ROLLOUT, beam=16))):
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
        res = engine.solve_residual(db, dm, state, **kw)
        engine.check_status(res["status"])
        st = state.cpu().numpy()
        for g in (1, 4):  # two of the graphs against the oracle's pure-Python solver
            n0, n1 = sl[g]
            adj, w = _scipy(hb, g), hb.weights[n0:n1]
            solver = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit, "rollout": orc.solve_mwis_rollout}[name]
            sel, _ = solver(fn, adj, w, 16) if name == "rollout" else solver(fn, adj, w)
            assert set(np.flatnonzero(st[n0:n1] == 1).tolist()) == set(int(v) for v in sel), (name, g)
