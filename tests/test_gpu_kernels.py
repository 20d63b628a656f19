"""Parity of the HIP kernels (through the C ABI) against the oracle.  Run on the GPU box: -m gpu.

Bars: integer / index / byte outputs bit-exact; float32 scores bit-exact against the CPU twin
(oracle/dgcn_oracle.c, same operation order) and within 1e-5 of the float64 restatement
(oracle/ref_numpy.py) - loosened to twice the float32 restatement's own distance from float64 on the
few ill-conditioned fixture graphs where plain float32 arithmetic itself exceeds 1e-5.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import TOL, check_scores
from distgcn_amd import _lib


def _dev(engine, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(engine.device)


def test_supports_bit_exact(engine, golden):
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    lap = engine.supports(db)
    lrp, lc, lv, fault = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    assert fault == 0
    assert np.array_equal(lap["row_ptr"].cpu().numpy(), lrp)
    assert np.array_equal(lap["col_idx"].cpu().numpy()[:lc.size], lc)
    got = lap["values"].cpu().numpy()[:lv.size]
    assert np.array_equal(got.view(np.uint32), lv.view(np.uint32))
    # and against the imported reference's simple_polynomials output (float64 -> float32 feed cast)
    import scipy.sparse as sp
    for i, (n0, n1) in enumerate(hb.graph_slices()):
        e0, e1 = lrp[n0], lrp[n1]
        m = sp.csr_matrix((got[e0:e1], lc[e0:e1] - n0, lrp[n0:n1 + 1] - e0), shape=(n1 - n0, n1 - n0))
        m.sort_indices()
        assert np.array_equal(m.indices, golden.supports["g%02d_lap_indices" % i])
        assert np.array_equal(m.data, golden.supports["g%02d_lap_data" % i].astype(np.float32))


def test_supports2_bit_exact(engine, golden):
    """T_2 = L.L on the device (dgcn_supports2_*: explicit product in SciPy's float64 order) against the CPU twin
    on every fixture graph and against the imported reference's simple_polynomials(adj, 2)[2], bit for bit."""
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    lap2 = engine.supports2(db)
    r2, c2, v2, fault = ctwin.supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    assert fault == 0
    assert np.array_equal(lap2["row_ptr"].cpu().numpy(), r2)
    assert np.array_equal(lap2["col_idx"].cpu().numpy()[:c2.size], c2)
    got = lap2["values"].cpu().numpy()[:v2.size]
    assert np.array_equal(got.view(np.uint32), v2.view(np.uint32))
    seen = 0
    for i, (n0, n1) in enumerate(hb.graph_slices()):
        if "g%02d_lap2_indptr" % i not in golden.supports.files:
            continue
        seen += 1
        e0, e1 = r2[n0], r2[n1]
        assert np.array_equal(r2[n0:n1 + 1] - e0, golden.supports["g%02d_lap2_indptr" % i])
        assert np.array_equal(c2[e0:e1] - n0, golden.supports["g%02d_lap2_indices" % i])
        assert np.array_equal(got[e0:e1], golden.supports["g%02d_lap2_data" % i].astype(np.float32))
    assert seen >= 3
    # ragged: empty graph, single vertex, edgeless graph, a star (hub row = every vertex), a 700-vertex graph
    import scipy.sparse as sp
    from distgcn_amd.batch import HostBatch
    star = sp.lil_matrix((130, 130)); star[0, 1:] = 1; star[1:, 0] = 1
    big = sp.random(700, 700, density=0.01, random_state=5, format="csr")
    big = ((big + big.T) > 0).astype(float); big.setdiag(0); big.eliminate_zeros()
    hb = HostBatch.from_scipy([sp.csr_matrix((0, 0)), sp.csr_matrix((1, 1)), sp.csr_matrix((5, 5)), sp.csr_matrix(star),
                               sp.csr_matrix(big)])
    db = engine.upload(hb)
    lap2 = engine.supports2(db)
    r2, c2, v2, fault = ctwin.supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    assert np.array_equal(lap2["row_ptr"].cpu().numpy(), r2)
    assert np.array_equal(lap2["col_idx"].cpu().numpy()[:c2.size], c2)
    assert np.array_equal(lap2["values"].cpu().numpy()[:v2.size].view(np.uint32), v2.view(np.uint32))


def test_supports_faults(engine):
    from distgcn_amd.batch import HostBatch
    from distgcn_amd._lib import DgcnError
    # self loop on vertex 1
    hb = HostBatch(np.array([0, 3]), np.array([0, 1, 3, 4]), np.array([1, 0, 1, 1]))
    db = engine.upload(hb)
    with pytest.raises(DgcnError, match="self-loop"):
        engine.supports(db)
    # column pointing into another graph
    hb = HostBatch(np.array([0, 2, 4]), np.array([0, 1, 2, 3, 4]), np.array([1, 0, 1, 2]))
    db = engine.upload(hb)
    with pytest.raises(DgcnError, match="column index"):
        engine.supports(db)


def test_spmm_split_rule_matches_twin(engine):
    """The split factor fixes the summation order: library and twin must agree for every width."""
    from oracle import ctwin
    for C in range(1, 257):
        assert engine.lib.dgcn_spmm_split(C) == ctwin.spmm_split(C), C


@pytest.mark.parametrize("C", [1, 2, 3, 4, 8, 16, 19, 32, 64])
@pytest.mark.parametrize("path", ["lds", "global"])
def test_spmm_bit_exact(engine, golden, C, path):
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    lap = engine.supports(db)
    lrp, lc, lv, _ = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    rng = np.random.default_rng(C)
    Z = rng.standard_normal((hb.num_nodes, C)).astype(np.float32)
    Y0 = rng.standard_normal((hb.num_nodes, C)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    kw = dict(graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs, max_nodes=hb.max_nodes) if path == "lds" else {}
    # plain SpMM (K4 alone)
    got = engine.spmm(lap, _dev(engine, Z), C, **kw).cpu().numpy()
    want = ctwin.spmm(lrp, lc, lv, Z, C)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # with the GraphConvolution epilogue (add_n, bias, leaky_relu)
    got = engine.spmm(lap, _dev(engine, Z), C, Y0=_dev(engine, Y0), ldy0=C, bias=_dev(engine, bias),
                      act="leaky_relu", **kw).cpu().numpy()
    want = ctwin.spmm(lrp, lc, lv, Z, C, Y0=Y0, bias=bias, act=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # independent check of the twin itself against SciPy float64
    import scipy.sparse as sp
    S = sp.csr_matrix((lv.astype(np.float64), lc, lrp), shape=(hb.num_nodes, hb.num_nodes))
    ref = S @ Z.astype(np.float64)
    assert np.abs(ctwin.spmm(lrp, lc, lv, Z, C) - ref).max() < 1e-4


@pytest.mark.parametrize("cin,ctot", [(32, 64), (16, 32), (64, 128), (32, 32), (1, 64), (32, 2), (19, 38), (5, 2)])
def test_transform_bit_exact(engine, cin, ctot):
    from oracle import ctwin
    rng = np.random.default_rng(cin * 1000 + ctot)
    for rows in (1, 31, 200, 4099):
        H = rng.standard_normal((rows, cin)).astype(np.float32)
        W = rng.standard_normal((cin, ctot)).astype(np.float32)
        got = engine.transform(_dev(engine, H), _dev(engine, W)).cpu().numpy()
        want = ctwin.transform(H, W)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rows, cin, ctot)
    # constant-feature form (H = NULL)
    W = rng.standard_normal((cin, ctot)).astype(np.float32)
    got = engine.transform(None, _dev(engine, W), rows=77, h_const=0.25).cpu().numpy()
    want = ctwin.transform(None, W, rows=77, h_const=0.25)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _modes():
    from distgcn_amd.engine import MODE_LAYERED, MODE_FUSED
    return [MODE_LAYERED, MODE_FUSED]


@pytest.mark.parametrize("mode", [0, 1])
def test_forward_golden_models(engine, golden, mode):
    """Every shipped-checkpoint fixture model on every fixture graph."""
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    ran = 0
    for mname in golden.model_names:
        layers = golden.layers(mname)
        got = engine.forward(db, DeviceModel(layers, engine.device), mode=mode).cpu().numpy()
        ran += 1
        twin = ctwin.forward(lap, layers, hb.num_nodes)
        assert np.array_equal(got.view(np.uint32), twin.view(np.uint32)), mname
        for i, (n0, n1) in enumerate(hb.graph_slices()):
            f64 = golden.scores["g%02d|%s|f64" % (i, mname)]
            f32 = golden.scores["g%02d|%s|f32" % (i, mname)]
            check_scores(got[n0:n1, 0], f32, f64, (mname, golden.names[i]))
    assert ran == len(golden.model_names)


def test_fused_shape_coverage(engine, golden, all_models):
    """Which shipped stacks the fused kernel takes: every [I, L] stack with hidden width <= 32 (narrower ones run
    zero-padded) and the two-layer stacks F -> c -> 1 with c = 48 / 64 (first layer in 32-column blocks); the two
    max_degree = 2 models run layer by layer and say so.  A deep stack with a 48-wide hidden layer is refused loudly."""
    from distgcn_amd import datagen
    from distgcn_amd._lib import DgcnError
    from distgcn_amd.engine import DeviceModel
    db = engine.upload(golden.host_batch([0]))
    wide = 0
    for name in all_models.names:
        layers = all_models.layers(name)
        dm = DeviceModel(layers, engine.device)
        widths = [lyr["weights"][0].shape[1] for lyr in layers[:-1]]
        # (the checkpoint named ld32_c32_l2 really holds a 48-wide hidden layer)
        ok = len(layers[0]["weights"]) == 2 and (all(w <= 32 for w in widths) or (len(layers) == 2 and widths[0] <= 128))
        wide += ok and any(w > 32 for w in widths)
        assert engine.solve_supported(db, dm) == ok, name
        if not ok:
            with pytest.raises(DgcnError, match="fused kernel handles|run layer by layer"):
                engine.forward(db, dm, mode=1)
    assert wide == 5  # c48 x 2 (one of them named c32), c64 x 3
    deep = DeviceModel(datagen.random_model(4, 48), engine.device)
    assert not engine.solve_supported(db, deep)
    with pytest.raises(DgcnError, match="fused kernel handles"):
        engine.forward(db, deep, mode=1)


@pytest.mark.parametrize("hidden", [33, 48, 64, 100, 128])
def test_fused_wide_two_layer_stacks(engine, golden, hidden):
    """F -> c -> 1 with 32 < c <= 128 in the fused kernel (first layer as 32-column blocks, the last layer's chains
    continued from block to block): scores bit-equal to the layer-by-layer path and to the twin, with a bias, with
    explicit features, on the residual-graph variant's plain launch too."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    for fs, bias in ((1, False), (16, True)):
        layers = datagen.random_model(2, hidden, feature_size=fs, bias=bias, seed=hidden + fs)
        dm = DeviceModel(layers, engine.device)
        assert engine.solve_supported(db, dm)
        X = None
        if fs > 1:
            X = np.random.default_rng(1).random((hb.num_nodes, fs)).astype(np.float32)
        Xd = None if X is None else _dev(engine, X)
        fused = engine.forward(db, dm, X=Xd, mode=1).cpu().numpy()
        layered = engine.forward(db, dm, X=Xd, mode=0).cpu().numpy()
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        twin = ctwin.forward(lap, layers, hb.num_nodes, X=X)
        assert np.array_equal(fused.view(np.uint32), layered.view(np.uint32))
        assert np.array_equal(fused.view(np.uint32), twin.view(np.uint32))
        if fs == 1:
            res = engine.solve(db, dm, mode=1)
            ref = ctwin.solve(hb, layers)
            assert np.array_equal(res["state"].cpu().numpy(), ref["state"])
            assert np.array_equal(res["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32))


@pytest.mark.parametrize("mode", [0, 1])
def test_forward_closed_form_l1(engine, golden, mode):
    """l=1, F=1: score_v = w0 + w1 * (1 - sum_u 1/sqrt(d_v d_u)); SURVEY anchors for ER_n200_p0.1_b0."""
    from distgcn_amd.engine import DeviceModel
    m = "result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn"
    hb = golden.host_batch([0])
    db = engine.upload(hb)
    got = engine.forward(db, DeviceModel(golden.layers(m), engine.device), mode=mode).cpu().numpy()[:, 0]
    assert np.allclose(got[:5], [0.6762156, 0.81670225, 0.65740633, 0.5908851, 0.7500816], atol=2e-7)
    p = golden.params(m)
    w0 = float(p["gcn_dqn/graphconvolution_1_vars/weights_0"][0, 0])
    w1 = float(p["gcn_dqn/graphconvolution_1_vars/weights_1"][0, 0])
    a = golden.scipy(0)
    d = np.asarray(a.sum(1)).ravel()
    dinv = np.where(d > 0, d ** -0.5, 0.0)
    closed = w0 + w1 * (1.0 - dinv * (a @ dinv))
    assert np.abs(got - closed).max() < TOL


@pytest.mark.parametrize("mode", [0, 1])
def test_forward_bias_relu_and_features(engine, golden, mode):
    """GCN2_DQN shape: bias on every layer, activation on the last layer too, explicit feature matrix."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    hb = golden.host_batch([1, 2, 8])
    db = engine.upload(hb)
    lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    layers = datagen.random_model(4, 32, feature_size=1, bias=True, last_act="leaky_relu", seed=3)
    rng = np.random.default_rng(5)
    X = (rng.random((hb.num_nodes, 1)) > 0.2).astype(np.float32)  # zero-weight rows -> 0 features
    got = engine.forward(db, DeviceModel(layers, engine.device), X=_dev(engine, X), mode=mode).cpu().numpy()
    want = ctwin.forward(lap, layers, hb.num_nodes, X=X)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode", [0, 1])
def test_forward_support_with_empty_rows(engine, mode):
    """A caller's support matrix may have rows without a single entry (here: L with its diagonal struck out, on graphs
    with 30 isolated vertices - two whole 16-row blocks of the fused kernel's aggregation are empty).  Such a row's output
    is act(Z0 + bias): both paths against the twin, bit for bit."""
    import ctypes as C
    from distgcn_amd import datagen, _lib
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(12)
    ps, cs, ws = [], [], []
    for n_conn, n_iso in ((40, 30), (100, 3), (17, 40)):
        indptr, indices = datagen.er_graph(n_conn, 0.2, rng)
        ps.append(np.concatenate([indptr, np.full(n_iso, indptr[-1], indptr.dtype)])); cs.append(indices)
        ws.append(np.ones(n_conn + n_iso))
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    db = engine.upload(hb)
    rp, ci, va = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
    rows = np.repeat(np.arange(hb.num_nodes), np.diff(rp))
    keep = ci != rows
    rp2 = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=hb.num_nodes))]).astype(np.int32)
    ci2, va2 = ci[keep].astype(np.int32), va[keep].astype(np.float32)
    assert (np.diff(rp2) == 0).sum() >= 73
    t = engine.torch
    d = {"row_ptr": t.from_numpy(rp2).to(engine.device), "col_idx": t.from_numpy(ci2).to(engine.device),
         "values": t.from_numpy(va2).to(engine.device)}
    per_graph = [int(rp2[n1] - rp2[n0]) for n0, n1 in hb.graph_slices()]
    d["c"] = _lib.DgcnCsr(hb.num_nodes, int(ci2.size), max(per_graph), d["row_ptr"].data_ptr(), d["col_idx"].data_ptr(), d["values"].data_ptr())
    db.lap = d
    layers = datagen.random_model(5, 32, bias=True, seed=13)
    got = engine.forward(db, DeviceModel(layers, engine.device), mode=mode).cpu().numpy()
    want = ctwin.forward((rp2, ci2, va2), layers, hb.num_nodes)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_row_longer_than_its_graph_is_a_fault_not_a_stray_write(engine):
    """The fused kernel's entry records live in a per-graph slice sized for rows of at most N entries.  A caller's matrix
    with a column repeated hundreds of times in one row (nothing validates that on the way in) must not write past the
    slice: the graph gets DGCN_FAULT_DEGREE_RANGE, its neighbours in the batch their usual results."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    rng = np.random.default_rng(3)
    good = datagen.er_batch(2, 64, 0.1, first_index=900)
    ps, cs, ws = [], [], []
    for n0, n1 in good.graph_slices():
        e0, e1 = int(good.row_ptr[n0]), int(good.row_ptr[n1])
        ps.append((good.row_ptr[n0:n1 + 1] - e0).astype(np.int64)); cs.append((good.col_idx[e0:e1] - n0).astype(np.int64)); ws.append(good.weights[n0:n1])
    n = 64
    rows = [[1] * 400] + [[0]] + [[] for _ in range(n - 2)]  # vertex 0: column 1, four hundred times
    ps.append(np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64))
    cs.append(np.array([c for r in rows for c in r], dtype=np.int64)); ws.append(np.ones(n))
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    dm = DeviceModel(datagen.random_model(6, 32, seed=2), engine.device)
    for order in (0, 1):  # the bad graph last, then first
        if order:
            hb = HostBatch.from_csr_lists(ps[::-1], cs[::-1], ws[::-1])
        res = engine.solve(engine.upload(hb), dm, mode=1)
        engine.torch.cuda.synchronize()
        assert int(res["status"].cpu().numpy()[0]) & 4
    alone = engine.solve(engine.upload(good), dm, mode=1)
    assert int(alone["status"].cpu().numpy()[0]) == 0


def _lgs_check(res, hb, golden, variant, ids):
    state = res["state"].cpu().numpy()
    rounds = res["rounds"].cpu().numpy()
    for j, i in enumerate(ids):
        n0, n1 = hb.graph_slices()[j]
        k = "g%02d_%s" % (i, variant)
        assert np.array_equal(np.flatnonzero(state[n0:n1] == 1), golden.lgs[k + "_set"]), k
        assert rounds[j] == golden.lgs[k + "_rounds"], k


@pytest.mark.parametrize("variant", ["raw", "ties1", "ties0", "signed", "int3"])
@pytest.mark.parametrize("lpv", [0, 1, 2, 4, 8])
def test_lgs_reference_goldens(engine, golden, variant, lpv, monkeypatch):
    """Sets / rounds / p2p / bst / overhead / totals equal to the imported reference's outputs."""
    ids = list(range(golden.num_graphs))
    hb = golden.host_batch(ids)
    db = engine.upload(hb)
    prio = np.concatenate([golden.lgs["g%02d_%s_prio" % (i, variant)] for i in ids])
    with _lib.options(lgs_lpv=lpv if lpv else -1):  # lanes per vertex: -1 = library default
        res = engine.lgs(db, prio=_dev(engine, prio), want_stats=True, want_overhead=True)
    engine.check_status(res["status"])
    _lgs_check(res, hb, golden, variant, ids)
    stats = res["stats"].cpu().numpy()
    oh = res["overhead"].cpu().numpy()
    tot = res["totals"].cpu().numpy()
    for j, i in enumerate(ids):
        n0, n1 = hb.graph_slices()[j]
        k = "g%02d_%s" % (i, variant)
        assert stats[j, 0] == golden.lgs[k + "_p2p"] and stats[j, 1] == golden.lgs[k + "_bst"], k
        assert np.array_equal(oh[n0:n1], golden.lgs[k + "_overhead"].astype(np.int32)), k
        assert tot[j] == pytest.approx(float(golden.lgs[k + "_total"]), rel=1e-12, abs=1e-12)
    # plain variant (no stats) must take the same decisions
    res2 = engine.lgs(db, prio=_dev(engine, prio))
    assert np.array_equal(res2["state"].cpu().numpy(), res["state"].cpu().numpy())
    assert np.array_equal(res2["rounds"].cpu().numpy(), res["rounds"].cpu().numpy())


@pytest.mark.parametrize("nstep", [1, 2])
def test_lgs_nstep(engine, golden, nstep):
    ids = list(range(golden.num_graphs))
    hb = golden.host_batch(ids)
    db = engine.upload(hb)
    for variant in ("raw", "ties1", "signed"):
        prio = np.concatenate([golden.lgs["g%02d_%s_prio" % (i, variant)] for i in ids])
        res = engine.lgs(db, prio=_dev(engine, prio), max_rounds=nstep)
        state = res["state"].cpu().numpy()
        for j, i in enumerate(ids):
            n0, n1 = hb.graph_slices()[j]
            k = "g%02d_%s_n%d" % (i, variant, nstep)
            assert np.array_equal(np.flatnonzero(state[n0:n1] == 1), golden.lgs[k + "_set"])
            assert np.array_equal(np.flatnonzero(state[n0:n1] == 2), golden.lgs[k + "_nb"])


def test_lgs_greedy_utility_pin(engine, golden):
    """greedy_utility stored by the reference in every .mat (Data_Generation.py:149-153, 218) pins the
    selection on raw weights: sequential greedy == local greedy for distinct weights."""
    hb = golden.host_batch()
    db = engine.upload(hb)
    res = engine.lgs(db, prio=db.weights, sum_weights=db.weights)
    tot = res["totals"].cpu().numpy()
    for i in range(golden.num_graphs):
        assert tot[i] == pytest.approx(float(golden.graphs["g%02d_greedy_utility" % i]), rel=1e-9)
        assert tot[i] == pytest.approx(float(golden.lgs["g%02d_raw_greedy_total" % i]), rel=1e-12)


def test_lgs_nan_fault_and_edge_cases(engine):
    import torch
    from distgcn_amd.batch import HostBatch
    from distgcn_amd._lib import DgcnError
    # graphs: empty graph, single vertex, two isolated vertices, a triangle
    hb = HostBatch(np.array([0, 0, 1, 3, 6]), np.array([0, 0, 0, 0, 2, 4, 6]), np.array([4, 5, 3, 5, 3, 4]))
    db = engine.upload(hb)
    prio = np.array([1.0, -1.0, float("-inf"), 0.5, 0.5, 0.5])
    res = engine.lgs(db, prio=_dev(engine, prio), want_stats=True)
    engine.check_status(res["status"])
    assert res["state"].cpu().numpy().tolist() == [1, 1, 1, 1, 2, 2]
    assert res["rounds"].cpu().numpy().tolist() == [0, 1, 1, 1]
    prio[4] = float("nan")
    res = engine.lgs(db, prio=_dev(engine, prio))
    with pytest.raises(DgcnError, match="NaN"):
        engine.check_status(res["status"])
    assert res["rounds"].cpu().numpy().tolist() == [0, 1, 1, -1]


def _independent_and_maximal(hb, state):
    rows = np.repeat(np.arange(hb.num_nodes), np.diff(hb.row_ptr))
    sel = state == 1
    assert not (sel[rows] & sel[hb.col_idx]).any(), "two adjacent vertices selected"
    has_sel_nb = np.zeros(hb.num_nodes, bool)
    has_sel_nb[rows[sel[hb.col_idx]]] = True
    assert (sel | has_sel_nb).all(), "set is not maximal"
    assert ((state == 2) == (~sel & has_sel_nb)).all()


@pytest.mark.parametrize("config", ["C2", "C3", "BA"])
@pytest.mark.parametrize("mode", [0, 1])
def test_solve_full_size_vs_twin(engine, golden, config, mode):
    """BASELINE.json full sizes with the TRAINED weights the benchmark runs (fixture copies of the shipped
    checkpoints): 500 ER graphs (C2: N=100, IS4SAT l=1; C3: N=200, IS4SAT l=20 c32) and a 500-graph BA test2 mix
    (C4's per-GPU share, DQNBA l=20).  Scores bit-exact vs the CPU twin, hence selected sets bit-identical; plus
    the size-independent properties (independence, maximality)."""
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    if config == "C2":
        hb, layers = datagen.er_batch(500, 100, 0.1), golden.layers("result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn")
    elif config == "C3":
        hb, layers = datagen.er_batch(500, 200, 0.1), golden.layers("result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn")
    else:
        hb, layers = datagen.ba_test2_batch(500), golden.layers("result_DQNBA_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn")
    db = engine.upload(hb)
    res = engine.solve(db, DeviceModel(layers, engine.device), mode=mode)
    engine.check_status(res["status"])
    ref = ctwin.solve(hb, layers)
    got = res["scores"].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref["scores"].view(np.uint32))
    state = res["state"].cpu().numpy()
    assert np.array_equal(state, ref["state"])
    assert np.array_equal(res["rounds"].cpu().numpy(), ref["rounds"])
    assert np.allclose(res["totals"].cpu().numpy(), ref["totals"], rtol=1e-12, atol=0)
    _independent_and_maximal(hb, state)


def test_argmax(engine, golden):
    hb = golden.host_batch()
    db = engine.upload(hb)
    rng = np.random.default_rng(0)
    s = rng.integers(0, 5, size=(hb.num_nodes, 1)).astype(np.float32)  # many ties: first maximum wins
    got = engine.argmax(db, _dev(engine, s)).cpu().numpy()
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        assert got[g] == int(np.argmax(s[n0:n1, 0]))


@pytest.mark.parametrize("mode", [0, 1])
def test_solve_edge_case_batch(engine, mode):
    """Ragged batch: an empty graph, a single vertex, an edgeless graph, a path, a 512-vertex sparse graph
    (the fused kernel's size limit) and a star whose hub has degree 300 - sets and scores vs the CPU twin."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(11)
    adjs = [sp.csr_matrix((0, 0)), sp.csr_matrix((1, 1)), sp.csr_matrix((7, 7))]
    path = sp.diags([np.ones(9), np.ones(9)], [1, -1], shape=(10, 10)).tocsr()
    adjs.append(path)
    big = sp.random(512, 512, density=0.004, random_state=3, format="csr")
    big = ((big + big.T) > 0).astype(float)
    big.setdiag(0)
    big.eliminate_zeros()
    adjs.append(sp.csr_matrix(big))
    star = sp.lil_matrix((301, 301))
    star[0, 1:] = 1
    star[1:, 0] = 1
    adjs.append(sp.csr_matrix(star))
    wts = [rng.random(a.shape[0]) for a in adjs]
    hb = HostBatch.from_scipy(adjs, wts)
    layers = datagen.random_model(5, 32, seed=4)
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    if mode == 1:
        assert engine.solve_supported(db, dm)
    res = engine.solve(db, dm, mode=mode)
    engine.check_status(res["status"])
    ref = ctwin.solve(hb, layers)
    assert np.array_equal(res["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32))
    assert np.array_equal(res["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(res["rounds"].cpu().numpy(), ref["rounds"])
    assert res["rounds"].cpu().numpy()[0] == 0 and res["totals"].cpu().numpy()[0] == 0.0
    # all-empty batch: nothing to do, nothing launched
    empty = HostBatch.from_scipy([sp.csr_matrix((0, 0))], [np.zeros(0)])
    r2 = engine.solve(engine.upload(empty), dm, mode=mode)
    assert r2["state"].numel() == 0


def test_every_shipped_checkpoint(engine, golden, all_models):
    """All 46 shipped checkpoints (hidden 1..64, F in {1, 2, 16, 32}, 1..20 layers, two with [I, L, L.L] supports)
    through the product path the agents use (fused when the shape allows, layer-by-layer otherwise): scores
    bit-equal to the twin, within 1e-5 of the float32 restatement (the closest available proxy of TF's float32
    path) and within max(1e-5, 2 |f32 - f64|) of the float64 one; sets equal to the oracle greedy run on the same
    priorities and - unless near-ties reorder priorities - to the reference's own local_greedy_search on the
    restatement.  Writes the per-model error table to gpurun_out/ (committed as profiles/r02_model_errors.txt)."""
    import os
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin, ref_numpy as orc
    assert len(all_models.names) == 46
    hb = golden.host_batch(all_models.graph_ids)
    db = engine.upload(hb)
    sups = [ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3], ctwin.supports2(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]]
    fused, same_as_reference, total = 0, 0, 0
    table = ["%-62s %5s %12s %12s %12s" % ("model (2 fixture graphs, %d vertices)" % hb.num_nodes, "path", "|hip-f32|", "|hip-f64|", "|f32-f64|")]
    for name in all_models.names:
        layers, meta = all_models.layers(name), all_models.meta(name)
        dm = DeviceModel(layers, engine.device)
        mode = 1 if engine.solve_supported(db, dm) else 0
        fused += mode
        res = engine.solve(db, dm, predict=meta["predict"], mode=mode)
        engine.check_status(res["status"])
        got = res["scores"].cpu().numpy().reshape(-1)
        state = res["state"].cpu().numpy()
        twin = ctwin.forward(sups[:meta["max_degree"]], layers, hb.num_nodes)[:, 0]
        assert np.array_equal(got.view(np.uint32), twin.view(np.uint32)), name
        prio = got.astype(np.float64) * hb.weights if meta["predict"] == "mwis" else got.astype(np.float64)
        e32 = e64 = e3264 = 0.0
        for gi, (n0, n1) in zip(all_models.graph_ids, hb.graph_slices()):
            f64, f32 = all_models.expect(gi, name, "f64"), all_models.expect(gi, name, "f32")
            a32, a64, a3264 = check_scores(got[n0:n1], f32, f64, (name, gi))
            e32, e64, e3264 = max(e32, a32), max(e64, a64), max(e3264, a3264)
            p, c, _ = golden.csr(gi)
            st, _ = orc.lgs_vectorised(p, c, prio[n0:n1])
            assert np.array_equal(state[n0:n1] == 1, st == 1), (name, gi)
            total += 1
            same_as_reference += set(np.flatnonzero(state[n0:n1] == 1)) == set(all_models.expect(gi, name, "set").tolist())
        table.append("%-62s %5s %12.3e %12.3e %12.3e" % (name, "fused" if mode else "layer", e32, e64, e3264))
    assert fused == 44  # every [I, L] stack: hidden width <= 32, or two layers with a wide first one (c48 / c64)
    assert same_as_reference == total, (same_as_reference, total)
    table.append("sets equal to the reference's local_greedy_search on the restatement's priorities: %d of %d" % (same_as_reference, total))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "model_errors.txt"), "w") as f:
        f.write("\n".join(table) + "\n")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_graphs_with_ties_all_paths(engine, seed):
    """Randomised sweep: 120 small graphs per seed (1..70 vertices, density 0..0.6, isolated vertices, weights
    drawn from a handful of values so ties are everywhere, some zero / negative): the greedy kernel, the
    layered solve and the fused solve against the twin and the vectorised oracle."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin, ref_numpy as orc
    rng = np.random.default_rng(1000 + seed)
    ps, cs, ws = [], [], []
    for _ in range(120):
        n = int(rng.integers(1, 71))
        p = float(rng.choice([0.0, 0.05, 0.2, 0.6]))
        indptr, indices = datagen.er_graph(n, p, rng)
        ps.append(indptr); cs.append(indices)
        ws.append(rng.choice([-1.0, 0.0, 0.25, 0.5, 0.5, 1.0, 2.0], size=n))
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    db = engine.upload(hb)
    # greedy kernel on the raw weights (ties broken by index)
    res = engine.lgs(db, prio=db.weights, sum_weights=db.weights)
    engine.check_status(res["status"])
    st = res["state"].cpu().numpy()
    rounds = res["rounds"].cpu().numpy()
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        want, r = orc.lgs_vectorised(ps[g], cs[g], ws[g])
        assert np.array_equal(st[n0:n1], want), g
        assert rounds[g] == r
    # whole path, both modes, positive weights (the agents' precondition)
    hb2 = HostBatch.from_csr_lists(ps, cs, [np.abs(w) + 0.125 for w in ws])
    db2 = engine.upload(hb2)
    layers = datagen.random_model(4, 32, seed=seed)
    ref = ctwin.solve(hb2, layers)
    for mode in (0, 1):
        out = engine.solve(db2, DeviceModel(layers, engine.device), mode=mode)
        engine.check_status(out["status"])
        assert np.array_equal(out["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32))
        assert np.array_equal(out["state"].cpu().numpy(), ref["state"])
        assert np.array_equal(out["rounds"].cpu().numpy(), ref["rounds"])
        assert np.allclose(out["totals"].cpu().numpy(), ref["totals"], rtol=1e-12, atol=0)


def test_large_batch_is_the_small_batch_tiled(engine):
    """Size-independent property at a batch far beyond BASELINE's: 10 000 graphs that are 20 copies of a
    500-graph batch must give 20 copies of its results (global node ids, int32 offsets, grid sizes)."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    base = datagen.er_batch(500, 200, 0.1)
    ps, cs, ws = [], [], []
    for g, (n0, n1) in enumerate(base.graph_slices()):
        e0, e1 = int(base.row_ptr[n0]), int(base.row_ptr[n1])
        ps.append(base.row_ptr[n0:n1 + 1] - e0)
        cs.append(base.col_idx[e0:e1] - n0)
        ws.append(base.weights[n0:n1])
    big = HostBatch.from_csr_lists(ps * 20, cs * 20, ws * 20)
    assert big.num_graphs == 10000 and big.num_nodes == 20 * base.num_nodes
    for mode, nl in ((1, 20), (0, 3)):
        dm = DeviceModel(datagen.random_model(nl, 32, seed=2), engine.device)
        small = engine.solve(engine.upload(base), dm, mode=mode)
        large = engine.solve(engine.upload(big), dm, mode=mode)
        engine.check_status(small["status"])
        engine.check_status(large["status"])
        for k in ("state", "rounds", "totals"):
            s, l = small[k].cpu().numpy(), large[k].cpu().numpy()
            assert np.array_equal(np.tile(s, 20), l), (mode, k)
        assert np.array_equal(np.tile(small["scores"].cpu().numpy().ravel().view(np.uint32), 20),
                              large["scores"].cpu().numpy().ravel().view(np.uint32))


def test_margin_risk_counts_and_guarantee(engine, golden):
    """dgcn_margin_risk_batch: counts equal the NumPy checker's on the fixture graphs for several
    deltas, and the guarantee holds - perturb every score by up to delta at random, solve again: every graph the kernel
    reported as risk-free keeps exactly its set."""
    from distgcn_amd.engine import DeviceModel
    from oracle import ref_numpy as orc
    hb = golden.host_batch()
    db = engine.upload(hb)
    dm = DeviceModel(golden.layers("result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn"), engine.device)
    res = engine.solve(db, dm, mode=1)
    engine.check_status(res["status"])
    scores = res["scores"].reshape(-1)
    state = res["state"].cpu().numpy()
    s_host = scores.cpu().numpy().astype(np.float64)
    prio = s_host * hb.weights
    rng = np.random.default_rng(5)
    some_risk = False
    for delta in (0.0, 2e-6, 2e-5, 1e-3, 1e-2):
        got = engine.margin_risk(db, res["state"], delta, scores=scores, weights=db.weights).cpu().numpy()
        for g, (n0, n1) in enumerate(hb.graph_slices()):
            p, c, _ = golden.csr(g)
            assert got[g] == orc.margin_risk(p, c, prio[n0:n1], state[n0:n1], delta, hb.weights[n0:n1]), (delta, g)
        some_risk |= bool(got.sum())
        for trial in range(3):
            pert = (s_host + rng.uniform(-delta, delta, size=s_host.size)) * hb.weights
            again = engine.lgs(db, prio=_dev(engine, pert), sum_weights=db.weights)["state"].cpu().numpy()
            for g, (n0, n1) in enumerate(hb.graph_slices()):
                if got[g] == 0:
                    assert np.array_equal(again[n0:n1] == 1, state[n0:n1] == 1), (delta, g, trial)
    assert some_risk  # the large deltas do flag graphs
    # prio given directly (|w| = 1) and an unweighted score run
    got = engine.margin_risk(db, res["state"], 1e-4, prio=_dev(engine, prio)).cpu().numpy()
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        p, c, _ = golden.csr(g)
        assert got[g] == orc.margin_risk(p, c, prio[n0:n1], state[n0:n1], 1e-4)


def test_layer_fused_with_next_transform(engine, golden):
    """Layer-by-layer path at hidden width 32: the aggregation of layer l and the transform of layer l + 1 run as one
    launch (csrc/layer.hip).  Same bits as the two separate kernels (option layer_fuse = 0) and as the twin, for deep stacks
    with and without bias, explicit features, the BA mix, and a batch whose largest graph (700 vertices) sends it back
    to the separate kernels."""
    import os
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    big = sp.random(700, 700, density=0.01, random_state=5, format="csr")
    big = ((big + big.T) > 0).astype(float); big.setdiag(0); big.eliminate_zeros()
    cases = [(golden.host_batch(), datagen.random_model(5, 32, seed=1), None),
             (datagen.ba_test2_batch(50), datagen.random_model(4, 32, bias=True, last_act="leaky_relu", seed=2), None),
             (golden.host_batch([0, 3]), datagen.random_model(3, 32, feature_size=8, seed=3), 8),
             (HostBatch.from_scipy([sp.csr_matrix(big), golden.scipy(0)]), datagen.random_model(4, 32, seed=4), None)]
    for hb, layers, fs in cases:
        db = engine.upload(hb)
        dm = DeviceModel(layers, engine.device)
        X = None if fs is None else np.random.default_rng(0).random((hb.num_nodes, fs)).astype(np.float32)
        Xd = None if X is None else _dev(engine, X)
        got = engine.forward(db, dm, X=Xd, mode=0).cpu().numpy()
        with _lib.options(layer_fuse=0):
            plain = engine.forward(db, dm, X=Xd, mode=0).cpu().numpy()
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        twin = ctwin.forward(lap, layers, hb.num_nodes, X=X)
        assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))
        assert np.array_equal(got.view(np.uint32), twin.view(np.uint32))
    # the launches really are fused: one "layer" launch per hidden layer boundary from layer 1 on
    hb, layers, _ = cases[0]
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    engine.supports(db)
    engine.timing(True)
    engine.forward(db, dm, mode=0)
    engine.torch.cuda.synchronize()
    engine.timing(False)
    # (layer 0's aggregation and layer 1's transform run apart - the two chains the contract carries in double - so a
    # 5-layer stack is: transform 0, precise spmm 0, precise transform 1, fused launches for layers 1..3, spmm 4)
    assert engine.timing_read("layer")[1] == 3 and engine.timing_read("transform")[1] == 2 and engine.timing_read("spmm")[1] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("layers_n", [2, 3, 20])
def test_cluster_variant_is_bit_identical(engine, golden, layers_n, monkeypatch, cluster_switch):
    """k_fused<.., CLUSTER>: one graph on K workgroups that hand their Z1 rows round through L2 every layer
    (csrc/fused.hip, "cluster variant").  Forced on for K = 2, 3, 4, 8 over ragged small batches - graph sizes that do
    not fill the last tile, single-tile graphs, an empty graph, graphs of 300 and 500 vertices (five to eight tiles per
    workgroup), biases, both activations - and compared bit for bit with
    the ordinary one-workgroup-per-graph launch (which the rest of this file pins against the CPU twin); also the
    automatic choice (small batch of N = 200 graphs) and repeated launches on the same buffers (progress words of
    earlier launches must not satisfy later ones)."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    rng = np.random.default_rng(5 + layers_n)
    layers = datagen.random_model(layers_n, 32)
    for i, lyr in enumerate(layers):  # biases everywhere, relu on odd layers
        lyr["bias"] = rng.uniform(-0.1, 0.1, np.asarray(lyr["weights"][0]).shape[1]).astype(np.float32)
        if i < layers_n - 1 and i % 2:
            lyr["act"] = "relu"
    model = DeviceModel(layers, engine.device)
    hbs = [datagen.er_batch(3, 200, 0.1, first_index=40), datagen.er_batch(1, 200, 0.1, first_index=41),
           datagen.er_batch(2, 500, 0.02, first_index=42),   # 32 tiles: K = 4 gives a workgroup eight (a tile per wave, two row sets)
           datagen.er_batch(1, 300, 0.05, first_index=43)]   # 19 tiles: K = 3 gives 7, 6, 6
    parts = [datagen.er_batch(1, n, p, first_index=50 + n) for n, p in ((137, 0.08), (16, 0.3), (33, 0.2), (512, 0.01), (64, 0.1))]
    ps, cs, ws = [np.zeros(1, np.int32)], [np.zeros(0, np.int32)], [np.zeros(0)]  # an empty graph first
    for hb in parts:
        ps.append(hb.row_ptr.astype(np.int32)); cs.append(hb.col_idx.astype(np.int32)); ws.append(hb.weights)
    hbs.append(HostBatch.from_csr_lists(ps, cs, ws))
    for hb in hbs:
        db = engine.upload(hb)
        ref = None
        for force in ("0", "2", "3", "4", "8", None):
            cluster_switch(force)
            out = engine.solve_buffers(db, True)
            for _ in range(3):
                engine.solve_fused(db, model, out=out, want_scores=True)
            torch.cuda.synchronize()
            got = {k: out[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")}
            assert int(got["status"][0]) == 0
            if ref is None:
                ref = got
                continue
            for k in ref:
                assert np.array_equal(ref[k].view(np.uint8), got[k].view(np.uint8)), (force, k, hb.num_graphs)


@pytest.mark.gpu
def test_cluster_variant_survives_foreign_use_of_its_buffers(engine, monkeypatch, cluster_switch):
    """The cluster variant's readers poll the exchange slices themselves (a chunk that still reads "unwritten" is not
    there yet), so what a previous launch - of any kind - left in the workspace must never look like data: every workgroup
    marks its rows at kernel start and the others wait for that once.  Alternate cluster launches of different batch
    shapes with ordinary launches that scribble over the same workspace, many times, automatic K (1, 8, 32 and 64
    graphs: the variant now runs whenever every workgroup is resident); always bit-identical to the ordinary launch."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.engine import DeviceModel
    model = DeviceModel(datagen.random_model(20, 32), engine.device)
    shapes = [(1, 200), (32, 200), (8, 150), (64, 200), (4, 300), (3, 500), (1, 200)]  # N = 500: entry values in global memory
    dbs = [engine.upload(datagen.er_batch(b, n, 0.1 if n <= 200 else (0.05 if n <= 300 else 0.02), first_index=900 + 7 * i))
           for i, (b, n) in enumerate(shapes)]
    big = engine.upload(datagen.er_batch(300, 120, 0.1, first_index=990))  # far too many graphs to cluster: an ordinary launch
    refs = []
    cluster_switch(0)
    for db in dbs:
        out = engine.solve_buffers(db, True)
        engine.solve_fused(db, model, out=out, want_scores=True)
        refs.append({k: out[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")})
    cluster_switch(None)
    for rep in range(4):
        for db, ref in zip(dbs, refs):
            out = engine.solve_buffers(db, True)
            for _ in range(1 + rep):
                engine.solve_fused(db, model, out=out, want_scores=True)
            got = {k: out[k].cpu().numpy().copy() for k in ref}
            assert int(got["status"][0]) == 0
            for k in ref:
                assert np.array_equal(ref[k].view(np.uint8), got[k].view(np.uint8)), (rep, k, db.num_graphs)
            engine.solve_fused(big, model, out=engine.solve_buffers(big, True))  # same workspace, other layout
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ba", "er300", "er308_hubs", "ragged", "two_layers", "features"])
def test_fused_words_in_global_variant_is_bit_identical(engine, case):
    """Round 6, C4: k_fused<.., GW> keeps only bufA / bufB / the row tables in LDS - entry values in global scratch, the 16-bit gather
    words built in bufA's space, written out while the layers run and read back into bufB's tail for the last layer and the greedy
    rounds - so that graphs of up to ~308 vertices leave TWO 512-thread workgroups per CU (three row blocks per wave beyond 256
    vertices) instead of one 1 024-thread workgroup per CU for every graph of a mixed batch.  Chosen by itself from three graphs
    per CU on; here forced (option fused_gw = 1) onto small batches and held against the ordinary launch (fused_gw = 0) and the
    twin, bit for bit: scores, states, rounds, totals; the BA mix, 300 vertices, 308 vertices with hub rows (rows longer than a
    tile's neighbours: the third row block), ragged sizes with empty / one-vertex graphs, a two-layer model, explicit features."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(61)
    layers = datagen.random_model(2 if case == "two_layers" else 6, 32, feature_size=3 if case == "features" else 1,
                                  bias=case != "ba", last_act="leaky_relu" if case == "ragged" else "linear", seed=17)
    if case == "ba":
        hb = datagen.ba_test2_batch(90, first_index=40)
    elif case == "er300":
        hb = datagen.er_batch(12, 300, 0.06, first_index=310)
    elif case == "er308_hubs":
        import scipy.sparse as sp
        mats, ws = [], []
        for i in range(5):
            ip, ix = datagen.er_graph(308 - i, 0.03, rng)
            a = sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(308 - i,) * 2).tolil()
            for v, d in ((0, 200), (17, 120), (307 - i, 250)):
                for u in rng.choice(np.setdiff1d(np.arange(308 - i), [v]), size=d, replace=False):
                    a[v, u] = 1.0
                    a[u, v] = 1.0
            a = a.tocsr(); a.sort_indices()
            mats.append(a); ws.append(rng.random(308 - i))
        hb = HostBatch.from_scipy(mats, ws)
    else:
        ps, cs, ws = [], [], []
        for n in ((0, 1, 2, 17, 64, 129, 257, 300, 305, 200, 16) if case == "ragged" else (257, 290, 120, 300)):
            if n == 0:
                ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                continue
            g = datagen.er_batch(1, n, min(0.9, 9.0 / max(n, 2)), first_index=int(rng.integers(1 << 20)))
            ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
        hb = HostBatch.from_csr_lists(ps, cs, ws)
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    assert engine.solve_supported(db, dm)
    X = None
    if case == "features":
        X = torch.from_numpy(rng.random((hb.num_nodes, 3)).astype(np.float32)).to(engine.device)
    got = {}
    for gw in (0, 1):
        with _lib.options(fused_gw=gw, fused_cluster=0):
            out = engine.solve_buffers(db, True)
            engine.solve_fused(db, dm, out=out, want_scores=True, X=X)
            got[gw] = engine.fetch_solve_buffers(out, hb.num_nodes, hb.num_graphs)
            assert got[gw]["status"] == 0, (case, gw)
    for k in ("state", "rounds"):
        assert np.array_equal(got[0][k], got[1][k]), (case, k)
    assert np.array_equal(got[0]["scores"].ravel().view(np.uint32), got[1]["scores"].ravel().view(np.uint32)), case
    assert np.array_equal(got[0]["totals"], got[1]["totals"]), case
    if X is None:
        want = ctwin.solve(hb, layers)
        assert np.array_equal(got[1]["scores"].ravel().view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), case
        assert np.array_equal(got[1]["state"], want["state"]) and np.array_equal(got[1]["rounds"], want["rounds"]), case
    # ... and the forward-only entry point (dgcn_gcn_forward_batch, mode 1) on the same variant
    if case in ("ba", "er308_hubs"):
        with _lib.options(fused_gw=1, fused_cluster=0):
            f1 = engine.forward(db, dm, mode=1).cpu().numpy()
        with _lib.options(fused_gw=0, fused_cluster=0):
            f0 = engine.forward(db, dm, mode=1).cpu().numpy()
        assert np.array_equal(f0.view(np.uint32), f1.view(np.uint32)), case
    # ... and the host-to-host object's compact transfer form, which the variant's image build reads as it is
    if case == "ba":
        from distgcn_amd.serving import HostSolver
        ps, cs, ws = [], [], []
        for n0, n1 in hb.graph_slices():
            e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
            ps.append((hb.row_ptr[n0:n1 + 1] - e0).astype(np.int32)); cs.append((hb.col_idx[e0:e1] - n0).astype(np.int32)); ws.append(hb.weights[n0:n1].copy())
        with _lib.options(fused_gw=1, fused_cluster=0, host_direct_bytes=0):
            hs = HostSolver(engine, dm, depth=1, want_scores=True)
            g = hs.solve(ps, cs, ws)
            hs.close()
        assert np.array_equal(g["state"], got[0]["state"]) and np.array_equal(g["rounds"], got[0]["rounds"])
        assert np.array_equal(g["scores"].view(np.uint32), got[0]["scores"].ravel().view(np.uint32))


def test_largest_first_dispatch_changes_nothing_but_the_order(engine, monkeypatch):
    """k_graph_rank (csrc/fused.hip): a mixed batch that needs several rounds of workgroups is dealt
    largest graph first.  Only the assignment of graphs to workgroups changes: scores, sets, rounds and totals are the
    same bits with the order forced on (ragged batch with empty and single-vertex graphs; the BA mix, where the library
    switches it on by itself) and off."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    parts = [datagen.er_batch(1, n, p, first_index=700 + n) for n, p in ((137, 0.08), (1, 0.5), (16, 0.3), (300, 0.05), (33, 0.2), (64, 0.1))]
    ps, cs, ws = [np.zeros(1, np.int32)], [np.zeros(0, np.int32)], [np.zeros(0)]  # an empty graph first
    for hb in parts * 3:
        ps.append(hb.row_ptr.astype(np.int32)); cs.append(hb.col_idx.astype(np.int32)); ws.append(hb.weights)
    ragged = HostBatch.from_csr_lists(ps, cs, ws)
    for layers_n, hb in ((20, datagen.ba_test2_batch(300)), (3, ragged), (1, ragged)):
        model = DeviceModel(datagen.random_model(layers_n, 32), engine.device)
        db = engine.upload(hb)
        got = {}
        for order in ("0", "1", None):
            with _lib.options(fused_order=-1 if order is None else int(order)):
                out = engine.solve_buffers(db, True)
                engine.solve_fused(db, model, out=out, want_scores=True)
                torch.cuda.synchronize()
            got[order] = {k: out[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")}
            assert int(got[order]["status"][0]) == 0
        for order in ("1", None):
            for k in got["0"]:
                assert np.array_equal(got["0"][k].view(np.uint8), got[order][k].view(np.uint8)), (layers_n, order, k)


def test_folded_dispatch_order_changes_nothing_but_the_order(engine):
    """Option fused_fold (csrc/fused.hip, fused_fold_at): in a launch whose workgroups are all resident at once, two per CU, the
    second half of the dispatch order runs smallest first, so that the largest graphs share their CUs with the smallest.  Still a
    permutation of the graphs - every graph solved exactly once, same bits - with the fold at the CU count (what the library
    chooses by itself for 257 .. 512 mixed graphs), in the middle of the batch, at its first and last position, and beyond it;
    mixed ER batch with an empty and a single-vertex graph, and the BA mix on k_fused<.., GW> (forced)."""
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    ps, cs, ws = [np.zeros(1, np.int32)], [np.zeros(0, np.int32)], [np.zeros(0)]
    for g in range(329):
        rng = np.random.default_rng(4100 + g)
        ip, ix = datagen.er_graph((80, 1, 120, 160, 200, 100, 140, 180)[g % 8], 0.1, rng)
        ps.append(ip); cs.append(ix); ws.append(rng.random(len(ip) - 1))
    mixed = HostBatch.from_csr_lists(ps, cs, ws)
    for hb, extra in ((mixed, {}), (datagen.ba_test2_batch(300), {"fused_gw": 1})):
        model = DeviceModel(datagen.random_model(6, 32), engine.device)
        db = engine.upload(hb)
        got = {}
        for fold in (0, -1, 1, 150, hb.num_graphs - 1, hb.num_graphs, 10 * hb.num_graphs):
            with _lib.options(fused_fold=fold, fused_order=1, fused_cluster=0, **extra):
                out = engine.solve_buffers(db, True)
                for k in ("state", "rounds"):
                    out[k].fill_(77)  # (a graph the permutation skipped would keep this)
                engine.solve_fused(db, model, out=out, want_scores=True)
                torch.cuda.synchronize()
            got[fold] = {k: out[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")}
            assert int(got[fold]["status"][0]) == 0
            for k in got[0]:
                assert np.array_equal(got[0][k].view(np.uint8), got[fold][k].view(np.uint8)), (fold, k)
        assert not (got[0]["rounds"] == 77).any()


@pytest.mark.parametrize("case", ["ties", "features_bias", "ba", "dense", "n512", "c2"])
def test_shallow_kernel_equals_fused_and_twin(engine, case, monkeypatch):
    """One-layer models go through the small dedicated kernel (csrc/shallow.hip: no 32-wide image, greedy rounds on the
    float64 priorities themselves instead of on ranks).  Against the twin bit for bit, and against k_fused forced on the
    same batch (option shallow = 0): ties everywhere (weights from a handful of values), isolated vertices, 1-vertex and
    empty graphs, explicit features with a bias and an activation (GCN2_DQN's last layer), the BA mix with hubs, dense
    graphs whose every row is long (with ties), 512-vertex graphs, and the C2 batch itself with the trained weights."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    rng = np.random.default_rng(77)
    X = None
    if case == "ties":
        ps, cs, ws = [], [], []
        for i in range(150):
            n = int(rng.integers(1, 71)) if i % 9 else 0
            indptr, indices = datagen.er_graph(n, float(rng.choice([0.0, 0.05, 0.2, 0.6])), rng) if n else (np.zeros(1, np.int64), np.zeros(0, np.int64))
            ps.append(indptr); cs.append(indices)
            ws.append(rng.choice([0.25, 0.5, 0.5, 1.0, 2.0], size=n))
        hb = HostBatch.from_csr_lists(ps, cs, ws)
        layers = datagen.random_model(1, 32, seed=3)
    elif case == "features_bias":
        hb = datagen.er_batch(40, 90, 0.08)
        layers = datagen.random_model(1, 32, feature_size=8, bias=True, last_act="leaky_relu", seed=4)
        X = rng.random((hb.num_nodes, 8)).astype(np.float32)
        X[rng.random(hb.num_nodes) < 0.1] = 0.0
    elif case == "ba":
        hb = datagen.ba_test2_batch(100)
        layers = datagen.random_model(1, 32, seed=5)
    elif case == "dense":  # every row long (40 - 110 entries on two or four lanes), explicit features: the gathering chain
        ps, cs, ws = [], [], []
        for n, p_ in ((300, 0.14), (200, 0.3), (160, 0.5), (260, 0.18), (129, 0.85)):  # (<= 14 000 entries: the shallow kernel's LDS)
            indptr, indices = datagen.er_graph(n, p_, rng)
            ps.append(indptr); cs.append(indices); ws.append(rng.choice([0.5, 1.0, 1.0, 2.0], size=n))
        hb = HostBatch.from_csr_lists(ps, cs, ws)
        layers = datagen.random_model(1, 32, feature_size=4, bias=True, last_act="relu", seed=8)
        X = rng.random((hb.num_nodes, 4)).astype(np.float32)
    elif case == "n512":
        hb = datagen.er_batch(6, 512, 0.02)
        layers = datagen.random_model(1, 32, bias=True, seed=6)
    else:
        hb = datagen.er_batch(500, 100, 0.1)
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models.npz"))
        from distgcn_amd.gcn.models import layers_from_params
        pre = "result_IS4SAT_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn|"
        layers = layers_from_params({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    db = engine.upload(hb)
    dm = DeviceModel(layers, engine.device)
    Xd = None if X is None else _dev(engine, X)
    if X is None:
        ref = ctwin.solve(hb, layers)
    else:
        lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
        sc = ctwin.forward(lap, layers, hb.num_nodes, X=X)
        ref = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, sc[:, 0].astype(np.float64) * hb.weights, sum_weights=hb.weights, want_stats=False)
        ref["scores"] = sc
    engine.timing(True)
    got = engine.solve(db, dm, mode=1, X=Xd)
    engine.torch.cuda.synchronize()
    engine.timing(False)
    engine.check_status(got["status"])
    with _lib.options(shallow=0):
        old = engine.solve(db, dm, mode=1, X=Xd)
        engine.check_status(old["status"])
    # graphs above 128 vertices have two shallow kernels - with and without the entry-parallel treatment of long rows (hubs) -
    # and the host picks by the batch's density: both, forced, on every case (the "ba" and "dense" cases have such rows)
    forced = []
    for flag in ("0", "1"):
        with _lib.options(shallow_long=int(flag)):
            out = engine.solve(db, dm, mode=1, X=Xd)
            engine.check_status(out["status"])
        forced.append(("shallow, option shallow_long=" + flag, out))
    for name, out in [("shallow", got), ("fused", old)] + forced:
        assert np.array_equal(out["scores"].cpu().numpy().reshape(-1).view(np.uint32), ref["scores"][:, 0].view(np.uint32)), name
        assert np.array_equal(out["state"].cpu().numpy(), ref["state"]), name
        assert np.array_equal(out["rounds"].cpu().numpy(), ref["rounds"]), name
        assert np.allclose(out["totals"].cpu().numpy(), ref["totals"], rtol=1e-12, atol=0), name
    # a self-loop is reported, not looped on
    bad = sp.csr_matrix(np.array([[1, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=float))
    hbb = HostBatch.from_scipy([bad], [np.ones(3)])
    r = engine.solve(engine.upload(hbb), dm if X is None else DeviceModel(datagen.random_model(1, 32), engine.device), mode=1)
    assert int(r["status"].cpu().numpy()[0]) & 1


def test_precise_kernels_bit_exact(engine, golden):
    """dgcn_spmm_f64acc_batch / dgcn_transform_f64acc_batch (the contract of layer index 0 / layer index 1: chains carried in
    double, rounded once) against the twin, and against a float64 SciPy / NumPy evaluation rounded to float32 where the
    two must agree exactly (a single rounding of a sum that float64 carries without error at these sizes is not
    guaranteed - so: within one float32 unit in the last place)."""
    import scipy.sparse as sp
    from oracle import ctwin
    hb = golden.host_batch()
    db = engine.upload(hb)
    lap = engine.supports(db)
    lrp, lc, lv, _ = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)
    rng = np.random.default_rng(12)
    for C in (1, 3, 32, 64):
        Z = rng.standard_normal((hb.num_nodes, C)).astype(np.float32)
        Y0 = rng.standard_normal((hb.num_nodes, C)).astype(np.float32)
        bias = rng.standard_normal(C).astype(np.float32)
        want = ctwin.spmm(lrp, lc, lv, Z, C, Y0=Y0, bias=bias, act=1, precise=True)
        for kw in (dict(), dict(graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs, max_nodes=hb.max_nodes)):  # plain kernel / LDS-staged
            got = engine.spmm(lap, _dev(engine, Z), C, Y0=_dev(engine, Y0), ldy0=C, bias=_dev(engine, bias), act="leaky_relu", precise=True, **kw).cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (C, bool(kw))
        got = engine.spmm(lap, _dev(engine, Z), C, precise=True).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ctwin.spmm(lrp, lc, lv, Z, C, precise=True).view(np.uint32)), C
        S = sp.csr_matrix((lv.astype(np.float64), lc, lrp), shape=(hb.num_nodes, hb.num_nodes))
        ref = (S @ Z.astype(np.float64)).astype(np.float32)
        assert np.abs(got - ref).max() <= np.spacing(np.abs(ref).max())
    for cin, ctot in ((32, 64), (16, 2), (64, 128), (5, 7)):
        H = rng.standard_normal((300, cin)).astype(np.float32)
        W = rng.standard_normal((cin, ctot)).astype(np.float32)
        got = engine.transform(_dev(engine, H), _dev(engine, W), precise=True).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ctwin.transform(H, W, precise=True).view(np.uint32)), (cin, ctot)
        ref = (H.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
        assert np.abs(got - ref).max() <= np.spacing(np.abs(ref).max())
