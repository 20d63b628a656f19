"""GPU: DGCN_RESIDUAL_FINISH_SMALL (csrc/tail.hip) - a graph with at most 64 undecided vertices runs the rest of its search
inside one launch.  A step is a function of the state alone and graphs are independent, so the final states must equal the
step-by-step kernels' (fused residual-graph kernel, any-size path) for every solver; checked from starts with decided
vertices, weightless graphs and graphs with nothing left, on graphs that enter the tail at once and on graphs that enter
it in the middle of their search, and against the oracle's solvers (mwis_gdpg_call.py:278-318, 343-384, 596-659)."""
import numpy as np
import pytest

from test_gpu_general import STEPPERS, _flags, _twin_scores_fn, general_switch  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _mixed_batch(golden):
    """fixture graphs (100 .. 300 vertices: they reach 64 undecided vertices in the middle of a search) plus ER graphs of 150, 300,
    64 and 65 vertices"""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    hb = golden.host_batch([2, 7, 1, 0, 12, 8])
    ps, cs, ws = [], [], []
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        r = hb.row_ptr[n0:n1 + 1]
        ps.append((r - r[0]).astype(np.int32)); cs.append((hb.col_idx[r[0]:r[-1]] - n0).astype(np.int32)); ws.append(hb.weights[n0:n1].copy())
    for n, p, seed in ((150, 0.05, 41), (300, 0.02, 42), (64, 0.1, 43), (65, 0.1, 44)):
        e = datagen.er_batch(1, n, p, first_index=seed)
        ps.append(e.row_ptr.astype(np.int32)); cs.append(e.col_idx.astype(np.int32)); ws.append(e.weights.copy())
    return HostBatch.from_csr_lists(ps, cs, ws)


@pytest.mark.parametrize("which", sorted(STEPPERS))
def test_finish_small_leaves_the_step_by_step_states(engine, golden, general_switch, which):
    import torch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    mode_name, max_rounds, (given, by_prio), predict = STEPPERS[which]
    agent = DQNAgent(_flags(num_layer=5, predict=predict), seed=9)
    rng = np.random.default_rng(5)
    for k in agent.model.vars:  # non-zero biases
        if k.endswith("/bias"):
            agent.model.vars[k] = rng.uniform(-0.2, 0.2, agent.model.vars[k].shape).astype(np.float32)
    agent.model._device_model = None
    hb = _mixed_batch(golden)
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
    if given:  # (the tail does not take given scores: the option bit must be ignored then)
        general_switch(None)
        full_scores = agent.model.forward_batch(engine, db, X=agent._features(hb), mode=1).clone()
    got = {}
    for path in (None, 1):
        for finish in (False, True):
            general_switch(path)
            s0 = torch.from_numpy(init.copy()).to(engine.device)
            res = engine.solve_residual(db, dm, s0, predict=predict, greedy=greedy, max_rounds=max_rounds, beam=6,
                                        weight_features=predict != "mwis", options=options, finish_small=finish,
                                        scores=None if full_scores is None else full_scores.clone())
            engine.check_status(res["status"])
            got[(path, finish)] = (s0.cpu().numpy().copy(), res["steps"])
    ref = got[(None, False)][0]
    for key, (st, steps) in got.items():
        assert np.array_equal(st, ref), (which, key, int((st != ref).sum()))
    if not given and got[(None, False)][1] > 3:  # fewer calls with the tail: the last steps of every graph are gone
        assert got[(None, True)][1] < got[(None, False)][1] and got[(1, True)][1] < got[(1, False)][1], (which, {k: v[1] for k, v in got.items()})
    assert np.array_equal(ref[sl[4][0]:sl[4][1]], init[sl[4][0]:sl[4][1]])  # the weightless graph was left alone


@pytest.mark.parametrize("which", ["dit", "cit", "rollout"])
def test_finish_small_against_the_oracle(engine, which):
    """the tail alone (graphs of at most 64 vertices: every step after the first call's runs there) against the oracle's solvers"""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    from oracle import ref_numpy as orc
    agent = DQNAgent(_flags(num_layer=4), seed=9)
    fn = _twin_scores_fn(agent.model.layers)
    ps, cs, ws = [], [], []
    for n, p, seed in ((30, 0.2, 1), (50, 0.1, 2), (64, 0.08, 3), (64, 0.3, 4), (17, 0.5, 5), (1, 0.5, 6), (40, 0.0, 7)):
        e = datagen.er_batch(1, n, p, first_index=900 + seed)
        ps.append(e.row_ptr.astype(np.int32)); cs.append(e.col_idx.astype(np.int32)); ws.append(e.weights.copy())
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
    greedy = {"dit": engine.GREEDY_ROUNDS, "cit": engine.GREEDY_CENTRAL, "rollout": engine.GREEDY_ROLLOUT}[which]
    res = engine.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=8, finish_small=True)
    engine.check_status(res["status"])
    assert res["steps"] <= 2  # the second call carries the option bit: its step, then the rest of every search in the tail launch
    st = res["state"].cpu().numpy()
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        adj = hb.scipy_graph(g).tocsr()
        adj.sort_indices()
        w = hb.weights[n0:n1]
        want = {"dit": orc.solve_mwis_dit, "cit": orc.solve_mwis_cit}[which](fn, adj, w) if which != "rollout" else orc.solve_mwis_rollout(fn, adj, w, b=8)
        assert set(int(v) for v in np.flatnonzero(st[n0:n1] == 1)) == want[0], (which, g)
        assert not np.any(st[n0:n1] == 0)


@pytest.mark.parametrize("which,family", [("rollout", "er"), ("cit", "er"), ("dit", "er"), ("rollout", "mc"), ("cit", "mc")])
def test_finish_small_at_search_size(engine, which, family):
    """C5-sized searches (ER N = 500, l = 20, b = 16) and joint 3 x 300 multi-channel graphs of 900 vertices (the any-size path): same final
    states with and without the tail, fewer calls with it; totals of the calls add up to the weight of the final sets."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.mwis_gdpg_call import DQNAgent
    agent = DQNAgent(_flags(num_layer=20), seed=3)
    if family == "er":
        hb = datagen.er_batch(6, 500, 0.02)
    else:
        import bench
        hb = bench.multichannel_batch(4, 300, 0.03, first_index=7)
    db = engine.upload(hb)
    dm = agent.model.device_model(engine)
    greedy = {"dit": engine.GREEDY_ROUNDS, "cit": engine.GREEDY_CENTRAL, "rollout": engine.GREEDY_ROLLOUT}[which]
    out = {}
    for finish in (False, True):
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=engine.device)
        res = engine.solve_residual(db, dm, state, greedy=greedy, max_rounds=1, beam=16, finish_small=finish)
        engine.check_status(res["status"])
        out[finish] = (state.cpu().numpy().copy(), res["steps"])
    assert np.array_equal(out[True][0], out[False][0]), (which, family, int((out[True][0] != out[False][0]).sum()))
    assert not np.any(out[True][0] == 0)
    if out[False][1] > 4:  # (dit decides most of a graph in its first rounds: three or four calls either way)
        assert out[True][1] < out[False][1], (out[True][1], out[False][1])
