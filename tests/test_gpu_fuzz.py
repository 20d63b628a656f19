"""Randomised parity: random batch shapes x model shapes through the fused path (ordinary launch, cluster variant,
largest-first dispatch, global-memory entry values - whatever the library picks for the shape) against the CPU twin,
bit for bit.  DGCN_FUZZ_CASES sets the number of cases per test (default 100: about a minute for the file; profiles/r05_fuzz_1000.txt,
r06_fuzz_1000.txt: 1 000 each)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_match_the_twin(engine):
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    cases = int(os.environ.get("DGCN_FUZZ_CASES", "100"))
    rng = np.random.default_rng(2026)
    for case in range(cases):
        layers_n = int(rng.choice([1, 2, 3, 8, 12, 20]))
        hidden = int(rng.choice([16, 32]))
        layers = datagen.random_model(layers_n, hidden, bias=bool(rng.integers(2)), seed=100 + case)
        kind = int(rng.integers(5))
        if kind == 4:  # many graphs of mixed sizes: largest first, and from 257 to 512 graphs the folded order (fused_fold_at)
            ps, cs, ws = [], [], []
            sizes = rng.choice([1, 7, 30, 64, 100, 150, 200], size=int(rng.integers(3, 6)), replace=False)
            for _ in range(int(rng.integers(257, 900))):
                n = int(rng.choice(sizes))
                ip, ix = datagen.er_graph(n, min(0.9, 6.0 / max(n, 2)), rng)
                ps.append(ip); cs.append(ix); ws.append(rng.random(n))
            hb = HostBatch.from_csr_lists(ps, cs, ws)
        elif kind == 0:  # a few graphs of one size: the cluster variant's ground
            n = int(rng.choice([60, 113, 150, 200, 257, 300, 400, 500, 512]))
            hb = datagen.er_batch(int(rng.integers(1, 12)), n, min(0.5, 12.0 / n), first_index=5000 + 50 * case)
        elif kind == 1:  # many graphs of one size
            n = int(rng.integers(1, 120))
            hb = datagen.er_batch(int(rng.integers(50, 400)), n, min(0.9, 6.0 / max(n, 2)), first_index=9000 + 500 * case)
        elif kind == 2:  # the BA mix: largest-first dispatch above 256 graphs
            hb = datagen.ba_test2_batch(int(rng.integers(20, 330)), first_index=13 * case)
        else:  # ragged, with empty and one-vertex graphs
            ps, cs, ws = [], [], []
            for _ in range(int(rng.integers(2, 40))):
                n = int(rng.choice([0, 1, 2, 17, 64, 129, 200, 333]))
                if n == 0:
                    ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                    continue
                g = datagen.er_batch(1, n, min(0.9, 8.0 / max(n, 2)), first_index=int(rng.integers(1 << 20)))
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
            hb = HostBatch.from_csr_lists(ps, cs, ws)
        if hb.num_nodes == 0:
            continue
        model = DeviceModel(layers, engine.device)
        db = engine.upload(hb)
        if not engine.solve_supported(db, model):
            continue
        out = engine.solve_buffers(db, True)
        engine.solve_fused(db, model, out=out, want_scores=True)
        got = engine.fetch_solve_buffers(out, hb.num_nodes, hb.num_graphs)
        assert got["status"] == 0, (case, got["status"])
        want = ctwin.solve(hb, layers)
        tag = (case, layers_n, hidden, kind, hb.num_graphs, hb.max_nodes)
        assert np.array_equal(got["scores"].ravel().view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), tag
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), tag
        assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), tag


def test_random_calls_through_the_host_solver_match_the_twin(engine):
    """The per-call path (one-slot HostSolver: packed batch read in place, completion word, cluster variant where it fits):
    random small batches, call after call on the same object, against the twin."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.serving import HostSolver
    from oracle import ctwin
    calls = int(os.environ.get("DGCN_FUZZ_CASES", "100")) * 4
    rng = np.random.default_rng(77)
    solvers = {}
    for call in range(calls):
        layers_n = int(rng.choice([1, 3, 12, 20]))
        if layers_n not in solvers:
            layers = datagen.random_model(layers_n, 32, bias=True, seed=layers_n)
            solvers[layers_n] = (layers, HostSolver(engine, DeviceModel(layers, engine.device), depth=1, want_scores=True))
        layers, hs = solvers[layers_n]
        ps, cs, ws = [], [], []
        for _ in range(int(rng.integers(1, 9))):
            n = int(rng.choice([0, 1, 30, 128, 200, 200, 200, 300, 500]))
            if n == 0:
                ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                continue
            g = datagen.er_batch(1, n, min(0.9, 10.0 / max(n, 2)), first_index=int(rng.integers(1 << 20)))
            ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
        hb = HostBatch.from_csr_lists(ps, cs, ws)
        got = hs.solve(ps, cs, ws)
        if hb.num_nodes == 0:
            assert got["state"].size == 0
            continue
        want = ctwin.solve(hb, layers)
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), (call, layers_n)
        assert np.array_equal(got["scores"].view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), (call, layers_n)
        assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), (call, layers_n)
    for _, hs in solvers.values():
        hs.close()


def _star(n, rng):
    """A hub with n - 1 leaves plus a few random edges between leaves: one row as long as the graph has vertices."""
    import scipy.sparse as sp
    rows, cols = [0] * (n - 1), list(range(1, n))
    for _ in range(n // 3):
        u, v = int(rng.integers(1, n)), int(rng.integers(1, n))
        if u != v:
            rows.append(u); cols.append(v)
    a = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    a = ((a + a.T) > 0).astype(np.float64).tocsr()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def test_random_shapes_down_the_any_size_path_match_the_twin(engine):
    """dgcn_solve_batch forced down the any-size path (csrc/general.hip, big.hip) on random shapes: k_big with 512 and 1 024
    threads (tile boundaries at 15 / 16 / 17 vertices, empty and one-vertex graphs, isolated vertices, hubs as long as their
    graph, graphs up to 976 vertices), explicit input features (the layer-by-layer front), wide and shallow models (the
    layer-by-layer kernels), graphs above 976 vertices - against the CPU twin, bit for bit."""
    from distgcn_amd import _lib, datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    lib = _lib.load()
    initial = int(lib.dgcn_get_general())
    cases = int(os.environ.get("DGCN_FUZZ_CASES", "100"))
    rng = np.random.default_rng(4040)
    try:
        lib.dgcn_set_general(1)
        for case in range(cases):
            layers_n = int(rng.choice([1, 2, 3, 4, 7, 20]))
            hidden = int(rng.choice([32, 32, 32, 16, 48]))
            fsize = int(rng.choice([1, 1, 2]))
            layers = datagen.random_model(layers_n, hidden, feature_size=fsize, bias=bool(rng.integers(2)), seed=300 + case)
            ps, cs, ws = [], [], []
            big = int(rng.choice([0, 0, 600, 976, 977, 1300]))
            for _ in range(int(rng.integers(1, 24))):
                n = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 64, 129, 255, 256, 257, 333, 512]))
                if n == 0:
                    ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                    continue
                if n >= 15 and rng.random() < 0.2:
                    ip, ix = _star(n, rng)
                    ps.append(ip); cs.append(ix); ws.append(rng.random(n))
                    continue
                g = datagen.er_batch(1, n, min(0.9, float(rng.choice([2.0, 8.0, 30.0])) / max(n, 2)), first_index=int(rng.integers(1 << 20)))
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
            if big:
                g = datagen.er_batch(1, big, 6.0 / big, first_index=7000 + case)
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
            hb = HostBatch.from_csr_lists(ps, cs, ws)
            if hb.num_nodes == 0:
                continue
            model = DeviceModel(layers, engine.device)
            db = engine.upload(hb)
            assert engine.solve_path(db, model) == 2
            X = None
            if rng.random() < 0.3:  # explicit features: layer 0 and the transform of layer 1 by the layer-by-layer kernels
                import torch
                X = torch.from_numpy(rng.random((hb.num_nodes, fsize)).astype(np.float32)).to(engine.device)
            out = engine.solve_buffers(db, True)
            engine.solve_fused(db, model, out=out, want_scores=True, X=X)
            got = engine.fetch_solve_buffers(out, hb.num_nodes, hb.num_graphs)
            assert got["status"] == 0, (case, got["status"])
            want = ctwin.solve(hb, layers) if X is None else None
            if X is not None:
                lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
                sc = ctwin.forward(lap, layers, hb.num_nodes, X=X.cpu().numpy())
                r = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, sc[:, 0].astype(np.float64) * hb.weights, sum_weights=hb.weights)
                want = {"scores": sc, "state": r["state"], "rounds": r["rounds"], "totals": r["totals"]}
            tag = (case, layers_n, hidden, fsize, X is not None, hb.num_graphs, hb.max_nodes)
            assert np.array_equal(got["scores"].ravel().view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), tag
            assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), tag
            assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), tag
    finally:
        lib.dgcn_set_general(initial)


def test_random_searches_with_and_without_the_tail(engine):
    """Iterative solvers on random batches (ragged sizes round the tail's 64-vertex limit, empty / one-vertex / edgeless graphs,
    stars, zero and missing weights, random starts), random model depth / bias / beam / solver variant, on the fused residual
    kernel and forced down the any-size path: the search with DGCN_RESIDUAL_FINISH_SMALL must leave the state bytes of the
    step-by-step search."""
    import torch
    from distgcn_amd import _lib, datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    lib = _lib.load()
    initial = int(lib.dgcn_get_general())
    cases = int(os.environ.get("DGCN_FUZZ_CASES", "100"))
    rng = np.random.default_rng(6464)
    try:
        for case in range(cases):
            # (1 / 2 layers: csrc/wide.hip's residual steps on the any-size path; 3+: k_big<RESID> / k_big2<RESID>)
            layers = datagen.random_model(int(rng.choice([1, 2, 3, 4, 6, 20])), 32, feature_size=int(rng.choice([1, 1, 3])),
                                          bias=bool(rng.integers(2)), seed=500 + case)
            ps, cs, ws = [], [], []
            if rng.random() < 0.15:  # now and then a graph beyond k_big's 976 vertices (k_big2 on the any-size path; the fused kernel
                n = int(rng.choice([980, 1100]))  # does not take the batch then: both "paths" are the any-size one)
                g = datagen.er_batch(1, n, 5.0 / n, first_index=int(rng.integers(1 << 20)))
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
            for _ in range(int(rng.integers(1, 10))):
                n = int(rng.choice([0, 1, 2, 16, 40, 63, 64, 65, 66, 100, 130, 200]))
                if n == 0:
                    ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                    continue
                if n >= 16 and rng.random() < 0.2:
                    ip, ix = _star(n, rng)
                    ps.append(ip); cs.append(ix); ws.append(rng.random(n))
                    continue
                g = datagen.er_batch(1, n, min(0.9, float(rng.choice([0.0, 2.0, 6.0, 20.0])) / max(n, 2)), first_index=int(rng.integers(1 << 20)))
                w = g.weights.copy()
                if rng.random() < 0.3:
                    w[rng.random(n) < 0.3] = 0.0
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(w)
            hb = HostBatch.from_csr_lists(ps, cs, ws)
            if hb.num_nodes == 0:
                continue
            model = DeviceModel(layers, engine.device)
            db = engine.upload(hb)
            greedy, max_rounds = [(engine.GREEDY_ROUNDS, 1), (engine.GREEDY_ROUNDS, 0), (engine.GREEDY_ROUNDS, 2), (engine.GREEDY_CENTRAL, 1),
                                  (engine.GREEDY_ROLLOUT, 1)][int(rng.integers(5))]
            options = engine.COMPLETE_BY_PRIORITY if rng.random() < 0.3 else 0
            predict = "mwis" if rng.random() < 0.7 else "mis"
            beam = int(rng.choice([1, 3, 16, 64]))
            init = np.where(rng.random(hb.num_nodes) < float(rng.choice([0.0, 0.2, 0.6])), rng.integers(1, 3, hb.num_nodes), 0).astype(np.uint8)
            got = {}
            for path in (-1, 1):
                lib.dgcn_set_general(path)
                for finish in (False, True):
                    s0 = torch.from_numpy(init.copy()).to(engine.device)
                    res = engine.solve_residual(db, model, s0, predict=predict, greedy=greedy, max_rounds=max_rounds, beam=beam,
                                                weight_features=predict != "mwis", options=options, finish_small=finish)
                    engine.check_status(res["status"])
                    got[(path, finish)] = s0.cpu().numpy().copy()
            ref = got[(-1, False)]
            for key, st in got.items():
                assert np.array_equal(st, ref), (case, key, len(layers), hb.num_graphs, hb.max_nodes, greedy, max_rounds, options, predict, beam,
                                                 int((st != ref).sum()))
    finally:
        lib.dgcn_set_general(initial)


def _hub_graph(n, hubs, p, rng):
    """ER(n, p) plus hubs [(vertex, degree), ..]: rows far longer than every other row of their sixteen-row tile."""
    import scipy.sparse as sp
    from distgcn_amd import datagen
    ip, ix = datagen.er_graph(n, p, rng)
    a = sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(n, n)).tolil()
    for v, d in hubs:
        others = np.setdiff1d(np.arange(n), [v])
        for u in rng.choice(others, size=min(d, n - 1), replace=False):
            a[v, u] = 1.0
            a[u, v] = 1.0
    a = a.tocsr()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def test_random_hub_rows_and_many_tile_graphs_match_the_twin(engine):
    """The class of round 4's silent wrong-sum bug (rows of 576+ entries in k_big's row-order counting sort) and of round 5's
    register-indexed tiles: random graphs of 520 .. 1 920 vertices - k_big up to 976, k_big2 with eight / twelve / FIFTEEN
    tiles per wave beyond - carrying hub rows of 576 .. n - 1 entries (several hubs, some sharing a tile, some at the tile's
    last row), next to small graphs in the same batch; deep stacks of width 32 and narrower (zero-padded onto the same
    kernels), biases, explicit input features now and then.  Plain solve against the CPU twin, bit for bit.
    (A quarter of DGCN_FUZZ_CASES cases: a 1 900-vertex twin forward is not free.)"""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    cases = max(1, int(os.environ.get("DGCN_FUZZ_CASES", "100")) // 4)
    rng = np.random.default_rng(57600)
    for case in range(cases):
        layers_n = int(rng.choice([3, 3, 4, 5, 8]))
        hidden = int(rng.choice([32, 32, 32, 16, 5]))
        fsize = int(rng.choice([1, 1, 1, 3]))
        layers = datagen.random_model(layers_n, hidden, feature_size=fsize, bias=bool(rng.integers(2)),
                                      last_act=str(rng.choice(["linear", "leaky_relu"])), seed=900 + case)
        ps, cs, ws = [], [], []
        n = int(rng.choice([520, 600, 640, 900, 976, 977, 1024, 1290, 1536, 1537, 1700, 1900, 1920]))
        nh = int(rng.integers(1, 4))
        hubs = []
        for h in range(nh):
            v = int(rng.choice([0, 15, 16, 17, n // 2, n - 17, n - 16, n - 1])) if rng.random() < 0.6 else int(rng.integers(n))
            if hubs and rng.random() < 0.4:
                v = min(n - 1, (hubs[0][0] // 16) * 16 + int(rng.integers(16)))  # a second hub in the first one's tile
            hubs.append((v, int(rng.integers(min(576, n // 2), n))))  # (520-vertex graphs: hubs of 260 .. 519 entries)
        ip, ix = _hub_graph(n, hubs, float(rng.choice([2.0, 6.0, 12.0])) / n, rng)
        ps.append(ip); cs.append(ix); ws.append(rng.random(n))
        for _ in range(int(rng.integers(0, 4))):  # company: small and medium graphs, one of them maybe without edges
            m = int(rng.choice([1, 16, 17, 200, 513, 700]))
            g = datagen.er_batch(1, m, min(0.9, float(rng.choice([0.0, 4.0, 10.0])) / max(m, 2)), first_index=int(rng.integers(1 << 20)))
            ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
        order = rng.permutation(len(ps))
        hb = HostBatch.from_csr_lists([ps[i] for i in order], [cs[i] for i in order], [ws[i] for i in order])
        model = DeviceModel(layers, engine.device)
        db = engine.upload(hb)
        assert engine.solve_path(db, model) == 2
        X = None
        if rng.random() < 0.25:
            import torch
            X = torch.from_numpy(rng.random((hb.num_nodes, fsize)).astype(np.float32)).to(engine.device)
        out = engine.solve_buffers(db, True)
        engine.solve_fused(db, model, out=out, want_scores=True, X=X)
        got = engine.fetch_solve_buffers(out, hb.num_nodes, hb.num_graphs)
        assert got["status"] == 0, (case, got["status"])
        if X is None:
            want = ctwin.solve(hb, layers)
        else:
            lap = ctwin.supports(hb.graph_ptr, hb.row_ptr, hb.col_idx)[:3]
            sc = ctwin.forward(lap, layers, hb.num_nodes, X=X.cpu().numpy())
            r = ctwin.lgs(hb.graph_ptr, hb.row_ptr, hb.col_idx, sc[:, 0].astype(np.float64) * hb.weights, sum_weights=hb.weights)
            want = {"scores": sc, "state": r["state"], "rounds": r["rounds"], "totals": r["totals"]}
        tag = (case, layers_n, hidden, fsize, X is not None, n, hubs, hb.num_graphs)
        assert np.array_equal(got["scores"].ravel().view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), tag
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), tag
        assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), tag
