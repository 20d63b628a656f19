"""CPU: batch packing, synthetic generators, sharding, flags, host mirrors of gcn.utils."""
import numpy as np
import pytest
import scipy.sparse as sp

from distgcn_amd import datagen, directory, parallel
from distgcn_amd.batch import HostBatch
from distgcn_amd.gcn import utils as gutils
from distgcn_amd.runtime_config import FLAGS, parse_argv


def test_batch_packing(golden):
    hb = golden.host_batch()
    assert hb.num_graphs == golden.num_graphs
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        a = hb.scipy_graph(g)
        assert (a != golden.scipy(g)).nnz == 0
        cols = hb.col_idx[hb.row_ptr[n0]:hb.row_ptr[n1]]
        assert cols.min(initial=n0) >= n0 and cols.max(initial=n0) < n1
    hb2 = HostBatch.from_scipy([sp.coo_matrix(golden.scipy(i)) for i in range(3)], [golden.csr(i)[2] for i in range(3)])
    assert np.array_equal(hb2.col_idx, golden.host_batch(range(3)).col_idx)
    sub = hb.subset(2, 5)
    assert sub.num_graphs == 3 and (sub.scipy_graph(0) != golden.scipy(2)).nnz == 0
    empty = HostBatch(np.array([0, 0, 2]), np.array([0, 0, 0]), np.zeros(0))
    assert empty.num_graphs == 2 and empty.max_nodes == 2 and empty.num_edges == 0
    with pytest.raises(ValueError):
        HostBatch(np.array([0, 2]), np.array([0, 1]), np.array([1]))


def test_generators_deterministic_and_well_formed():
    a = datagen.er_batch(6, 50, 0.1)
    b = datagen.er_batch(6, 50, 0.1)
    assert np.array_equal(a.col_idx, b.col_idx) and np.array_equal(a.weights, b.weights)
    c = datagen.er_batch(3, 50, 0.1, first_index=3)
    assert np.array_equal(c.weights, a.weights[150:])
    for hb in (a, datagen.ba_test2_batch(10)):
        for g in range(hb.num_graphs):
            m = hb.scipy_graph(g)
            assert (m != m.T).nnz == 0 and m.diagonal().sum() == 0
            assert np.all(np.diff(m.indptr) >= 0)
    ba = datagen.ba_test2_batch(25)
    sizes = np.diff(ba.graph_ptr)
    assert sorted(set(sizes.tolist())) == [100, 150, 200, 250, 300]
    m = ba.scipy_graph(4)  # N=100, m=20: (N - m - 1) * m + m edges
    assert m.nnz // 2 == (100 - 21) * 20 + 20


def test_multichannel_batch_generator_is_seeded_and_well_formed():
    """datagen.multichannel_batch (the MC900 / MC900-l1 / MC1500 bench and parity configurations): K per-channel copies of a
    seeded conflict graph joined as wireless_rollout_test_flood.py:98-133 joins them - K * nflows vertices, symmetric, no
    self-loops, every flow's copies a clique across the channels; the same graphs for the same index whatever the batch
    they are drawn in (a rank of a sharded job generates its own range)."""
    from distgcn_amd import datagen
    a = datagen.multichannel_batch(3, 40, 0.1, first_index=5)
    b = datagen.multichannel_batch(1, 40, 0.1, first_index=6)
    assert a.num_graphs == 3 and a.max_nodes == 120 and np.all(np.diff(a.graph_ptr) == 120)
    g1 = a.scipy_graph(1).tocsr()
    assert (g1 != b.scipy_graph(0).tocsr()).nnz == 0 and np.array_equal(a.weights[120:240], b.weights)
    assert g1.diagonal().sum() == 0 and (g1 != g1.T).nnz == 0
    dense = g1.toarray()
    for f in range(40):  # the copies of flow f on the three channels conflict with each other
        for c1 in range(3):
            for c2 in range(3):
                if c1 != c2:
                    assert dense[c1 * 40 + f, c2 * 40 + f] == 1.0
    # nothing links different flows across channels
    for c1 in range(3):
        for c2 in range(3):
            if c1 != c2:
                blk = dense[c1 * 40:(c1 + 1) * 40, c2 * 40:(c2 + 1) * 40]
                assert np.array_equal(blk, np.eye(40))


def test_shard_ranges_cover_and_balance():
    hb = datagen.ba_test2_batch(50)
    for world in (1, 2, 3, 8):
        r = parallel.shard_ranges(hb, world)
        assert r[0][0] == 0 and r[-1][1] == hb.num_graphs
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    r = parallel.shard_ranges(hb, 4)
    cost = [int(hb.graph_ptr[b] - hb.graph_ptr[a] + hb.row_ptr[hb.graph_ptr[b]] - hb.row_ptr[hb.graph_ptr[a]]) for a, b in r]
    assert max(cost) < 1.6 * (sum(cost) / 4)


def test_flags_and_directory():
    assert FLAGS.num_layer == 20 and FLAGS.predict == "mwis" and FLAGS.max_degree == 1
    fl = FLAGS.copy(training_set="IS4SAT", feature_size=1, hidden1=32, num_layer=20, max_degree=1, diver_num=1)
    assert directory.find_model_folder(fl, "dqn") == "./model/result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn"
    saved = FLAGS.hidden1
    rest = parse_argv(["--hidden1=16", "--unknown=3", "pos"])
    assert FLAGS.hidden1 == 16 and rest == ["--unknown=3", "pos"]
    FLAGS.hidden1 = saved


def test_host_utils_match_goldens(golden):
    adj = golden.scipy(3)
    sup = gutils.simple_polynomials(adj, 1)
    lap = sp.csr_matrix((sup[1][1], (sup[1][0][:, 0], sup[1][0][:, 1])), shape=sup[1][2])
    lap.sort_indices()
    assert np.array_equal(lap.data, golden.supports["g03_lap_data"])
    w = golden.csr(3)[2]
    f = gutils.preprocess_features(sp.lil_matrix(np.ones([w.size, 1]) * w[:, None]))
    dense = np.zeros(w.size)
    dense[f[0][:, 0]] = f[1]
    assert np.array_equal(dense, golden.supports["g03_feat_rownorm"])


def _few_big_many_small():
    rng = np.random.default_rng(4)
    ps, cs, ws = [], [], []
    for n in [60] * 160 + [300] * 40:
        p, c = datagen.er_graph(n, 0.1, rng)
        ps.append(p); cs.append(c); ws.append(rng.random(n))
    return HostBatch.from_csr_lists(ps, cs, ws)


def test_size_buckets_and_select():
    assert len(datagen.ba_test2_batch(100).size_buckets()) == 1  # 44 % large graphs: one launch is faster
    hb = _few_big_many_small()
    buckets = hb.size_buckets()
    assert len(buckets) == 2 and sorted(np.concatenate(buckets).tolist()) == list(range(200))
    sub = hb.select(buckets[1])
    assert sub.num_graphs == buckets[1].size
    g = int(buckets[1][3])
    assert (sub.scipy_graph(3) != hb.scipy_graph(g)).nnz == 0
    n0, n1 = hb.graph_slices()[g]
    assert np.array_equal(sub.weights[sub.graph_ptr[3]:sub.graph_ptr[4]], hb.weights[n0:n1])
    assert len(datagen.er_batch(40, 100, 0.1).size_buckets()) == 1


def test_mat_loader_when_reference_available():
    import os
    from distgcn_amd import harness
    path = "/root/reference/data/ER_Graph_Uniform_GEN21_test2"
    if not os.path.isdir(path):
        pytest.skip("reference data not present (GPU box)")
    d = harness.load_mat_folder(path, limit=3)
    assert len(d["adjs"]) == 3 and d["adjs"][0].shape[0] == d["weights"][0].size
    assert d["greedy_utility"][0] > 0 and d["adjs"][0].format == "csr"


def test_wireless_traffic_follows_reference_draws():
    """make_traffic draws what wireless_dqn_test.py:181-199 draws after np.random.seed(seed)."""
    from distgcn_amd import wireless
    tr = wireless.make_traffic(7, 20, 0.05, seed=3)
    np.random.seed(3)
    arrival_rate = 0.5 * (0 + 100) * 0.05
    inter = np.random.exponential(1.0 / arrival_rate, (7, int(2 * 20 * arrival_rate)))
    at = np.cumsum(inter, axis=1)
    acc = np.zeros((7, 20))
    for t in range(20):
        acc[:, t] = np.count_nonzero(at < t, axis=1)
    lr = np.random.normal(50, 25, size=[20, 7, 1]).astype(int)
    lr[lr < 0] = 0
    lr[lr > 100] = 100
    assert np.array_equal(np.diff(acc, prepend=0).transpose(), tr["arrival_pkts"])
    assert np.array_equal(lr, tr["link_rates"])
    # the restated slot loop conserves packets: arrivals = departures + backlog
    from oracle import ref_wireless
    import scipy.sparse as sp
    adj = sp.csr_matrix(np.triu(np.random.RandomState(1).rand(7, 7) < 0.3, 1).astype(float))
    adj = adj + adj.T
    res = ref_wireless.simulate_one(adj, tr["arrival_pkts"], tr["link_rates"],
                                    lambda a, w: set(np.flatnonzero(w == w.max()).tolist()[:1]), "qr")
    assert np.isclose(tr["arrival_pkts"][1:].sum(), res["depart"].sum() + res["queue"][-1].sum())


@pytest.mark.parametrize("compress", [False, True])
def test_matfile_reader_round_trip(tmp_path, compress):
    """distgcn_amd.matfile against files written by scipy.io.savemat the way Data_Generation.py:218-219 does."""
    import scipy.io as sio
    from distgcn_amd import harness, matfile
    rng = np.random.default_rng(2)
    for i, n in enumerate((1, 7, 120)):
        p, c = datagen.er_graph(n, 0.2, rng)
        adj = sp.csr_matrix((np.ones(c.size), c, p), shape=(n, n))
        w = rng.random((1, n))
        sio.savemat(str(tmp_path / ("g%d.mat" % i)), {"adj": adj.tocsc(), "weights": w, "N": n, "p": 0.2,
                                                      "greedy_utility": 1.5 + i, "mwis_utility": 2.5, "mwis_label": (w > 0.5) * 1,
                                                      "tag": "abc"}, do_compression=compress)
        m = matfile.loadmat(str(tmp_path / ("g%d.mat" % i)))
        ref = sio.loadmat(str(tmp_path / ("g%d.mat" % i)))
        ip, ix = matfile.symmetric_csr(m["adj"], strict=True)
        assert np.array_equal(ip, adj.indptr) and np.array_equal(ix, adj.indices)
        assert m["adj"].data is None or np.array_equal(m["adj"].data, np.ones(c.size))
        for k in ("weights", "N", "p", "greedy_utility", "mwis_utility", "mwis_label"):
            assert np.array_equal(np.asarray(m[k]), ref[k]) and np.asarray(m[k]).shape == ref[k].shape, k
        assert m["tag"] == "abc"
    data = harness.load_mat_folder(str(tmp_path))
    assert data["names"] == ["g0.mat", "g1.mat", "g2.mat"] and data["greedy_utility"] == [1.5, 2.5, 3.5]
    assert data["adjs"][2].shape == (120, 120) and (data["adjs"][2] != adj).nnz == 0
    # an asymmetric matrix is refused by the fast path and read through SciPy instead
    bad = sp.csr_matrix(np.triu(np.ones((4, 4)), 1))
    with pytest.raises(ValueError):
        matfile.symmetric_csr(matfile.SparseCSC((4, 4), bad.tocsc().indptr, bad.tocsc().indices, None))


def _random_csr_lists(rng, count, dtype):
    from distgcn_amd import datagen
    ps, cs, ws = [], [], []
    for _ in range(count):
        n = int(rng.integers(0, 40))
        ip, ix = datagen.er_graph(n, float(rng.choice([0.0, 0.1, 0.5])), rng) if n else (np.zeros(1, np.int64), np.zeros(0, np.int64))
        ps.append(ip.astype(dtype)); cs.append(ix.astype(dtype)); ws.append(rng.random(n))
    return ps, cs, ws


def _numpy_pack(ps, cs, ws):
    """The block-diagonal layout written out in NumPy (what dgcn_pack_batch must produce)."""
    sizes = np.array([p.size - 1 for p in ps], dtype=np.int64)
    nnz = np.array([p[-1] for p in ps], dtype=np.int64)
    gp = np.concatenate([[0], np.cumsum(sizes)])
    ep = np.concatenate([[0], np.cumsum(nnz)])
    rp = np.concatenate([p[:-1].astype(np.int64) + ep[g] for g, p in enumerate(ps)] + [[ep[-1]]]) if ps else np.zeros(1)
    ci = np.concatenate([c.astype(np.int64) + gp[g] for g, c in enumerate(cs)]) if ps else np.zeros(0)
    return gp.astype(np.int32), rp.astype(np.int32), ci.astype(np.int32), (np.concatenate(ws) if ps else np.zeros(0))


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_native_pack_matches_numpy(dtype):
    """dgcn_pack_batch (host C++ in libdgcn.so, bound through the C ABI) against the layout written out in NumPy:
    ragged batch with empty and edgeless graphs, both index widths, with and without weights, 1..8 threads."""
    from distgcn_amd.batch import HostBatch, pack_csr_lists
    rng = np.random.default_rng(3)
    ps, cs, ws = _random_csr_lists(rng, 57, dtype)
    gp, rp, ci, w = _numpy_pack(ps, cs, ws)
    for threads in (1, 3, 8):
        hb = HostBatch.from_packed(*pack_csr_lists(ps, cs, ws, threads=threads))
        assert np.array_equal(hb.graph_ptr, gp) and np.array_equal(hb.row_ptr, rp) and np.array_equal(hb.col_idx, ci)
        assert np.array_equal(hb.weights, w)
        assert hb.max_nodes == max(p.size - 1 for p in ps) and hb.max_graph_edges == max(int(p[-1]) for p in ps)
        assert hb.max_degree == max(int(np.diff(p).max()) if p.size > 1 else 0 for p in ps)
    hb = HostBatch.from_csr_lists(ps, cs)  # no weights
    assert hb.weights is None and np.array_equal(hb.col_idx, ci)
    # mixed index widths fall back to the NumPy path and give the same batch
    mixed = HostBatch.from_csr_lists([p.astype(np.int64) for p in ps], [c.astype(np.int32) for c in cs], ws)
    assert np.array_equal(mixed.row_ptr, rp) and np.array_equal(mixed.col_idx, ci) and mixed.max_degree == hb.max_degree
    # a larger batch takes several worker threads
    ps, cs, ws = _random_csr_lists(np.random.default_rng(4), 900, dtype)
    gp, rp, ci, w = _numpy_pack(ps, cs, ws)
    hb = HostBatch.from_csr_lists(ps, cs, ws)
    assert np.array_equal(hb.graph_ptr, gp) and np.array_equal(hb.row_ptr, rp) and np.array_equal(hb.col_idx, ci)


def test_native_pack_rejects_malformed_input():
    from distgcn_amd._lib import DgcnError
    from distgcn_amd.batch import HostBatch, pack_csr_lists
    i32 = lambda *a: np.array(a, dtype=np.int32)
    with pytest.raises(DgcnError, match="non-decreasing"):
        HostBatch.from_csr_lists([i32(0, 2, 1, 3)], [i32(1, 2, 0)])
    with pytest.raises(DgcnError, match="outside"):
        HostBatch.from_csr_lists([i32(0, 1, 2)], [i32(1, 7)])
    with pytest.raises(DgcnError, match="start at 0"):
        HostBatch.from_csr_lists([i32(1, 1, 2)], [i32(1, 0)])
    with pytest.raises(ValueError, match="does not match its indptr"):
        HostBatch.from_csr_lists([i32(0, 1, 2)], [i32(1)])
    with pytest.raises(ValueError, match="vertex count"):
        HostBatch.from_csr_lists([i32(0, 1, 2)], [i32(1, 0)], [np.ones(3)])
    with pytest.raises(ValueError, match="too small"):
        pack_csr_lists([i32(0, 1, 2)], [i32(1, 0)], [np.ones(2)], staging=np.empty(8, np.uint8))
    empty = HostBatch.from_csr_lists([], [], [])
    assert empty.num_graphs == 0 and empty.num_nodes == 0 and empty.graph_ptr.tolist() == [0]


def test_native_pack_survives_fork():
    """The packer's worker threads do not exist in a forked child: it must start its own instead of waiting for them."""
    import os
    from distgcn_amd.batch import HostBatch
    ps, cs, ws = _random_csr_lists(np.random.default_rng(9), 900, np.int32)
    want = HostBatch.from_csr_lists(ps, cs, ws)  # parent: several workers started
    pid = os.fork()
    if pid == 0:
        ok = 1
        try:
            got = HostBatch.from_csr_lists(ps, cs, ws)
            ok = 0 if (np.array_equal(got.col_idx, want.col_idx) and np.array_equal(got.row_ptr, want.row_ptr)) else 2
        finally:
            os._exit(ok)
    import time
    t0 = time.time()
    while time.time() - t0 < 60:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
            return
        time.sleep(0.05)
    os.kill(pid, 9)
    raise AssertionError("forked child hung in dgcn_pack_batch")


def test_one_call_helper_checks_its_arguments_before_calling_anything():
    """csrc/pyptr.c solve_lists(): the per-graph API's one interpreter call around dgcn_host_solver_submit/_result.  Its
    argument checks run before either entry point is touched (the addresses below are 0), with the interpreter path's
    error types: TypeError for index / weight dtypes the native packer does not take (callers fall back on those),
    ValueError for lengths that do not fit together."""
    from distgcn_amd.batch import _pyptr
    m = _pyptr()
    assert m is not None and hasattr(m, "solve_lists"), "the CPython helper has not been built (__graft_entry__.build)"
    ip, ix, w = np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), np.ones(2)
    with pytest.raises(ValueError, match="indices array does not match"):
        m.solve_lists(0, 0, 0, [ip], [ix[:1]], [w])
    with pytest.raises(ValueError, match="weights array does not match"):
        m.solve_lists(0, 0, 0, [ip], [ix], [np.ones(3)])
    with pytest.raises(ValueError, match="differ in length"):
        m.solve_lists(0, 0, 0, [ip, ip], [ix], None)
    with pytest.raises(ValueError, match="empty"):
        m.solve_lists(0, 0, 0, [np.zeros(0, np.int32)], [np.zeros(0, np.int32)], None)
    with pytest.raises(TypeError, match="int32 or int64"):
        m.solve_lists(0, 0, 0, [ip.astype(np.int64)], [ix], [w])  # two widths in one call
    with pytest.raises(TypeError, match="int32 or int64"):
        m.solve_lists(0, 0, 0, [ip.astype(np.float64)], [ix], [w])
    with pytest.raises(TypeError, match="float64"):
        m.solve_lists(0, 0, 0, [ip], [ix], [w.astype(np.float32)])
    with pytest.raises((ValueError, BufferError)):
        m.solve_lists(0, 0, 0, [np.arange(6, dtype=np.int32)[::2]], [ix], [w])  # not contiguous
    with pytest.raises(ValueError, match="more than 64"):
        m.solve_lists(0, 0, 0, [ip] * 65, [ix] * 65, None)


def test_multichannel_conflict_graphs_match_the_reference(golden):
    """wireless.multichannel_conflict_simulate / multichannel_conflict_graph against what the reference's own functions
    returned (tests/golden/multichannel.npz: wireless_rollout_test_flood.py:70-95 and :98-133 cut out with ast and
    executed by oracle/make_golden_multichannel.py): same per-channel graphs from the same np.random seed, same joint
    graph on K * nn vertices (diagonal blocks + per-link cliques), entry for entry."""
    import os
    import scipy.sparse as sp
    from conftest import GOLDEN
    from distgcn_amd import wireless
    z = np.load(os.path.join(GOLDEN, "multichannel.npz"))
    for ci, (gi, k, p, seed) in enumerate(z["cases"]):
        gi, k, seed = int(gi), int(k), int(seed)
        adj = golden.scipy(gi)
        np.random.seed(seed)
        graphs = wireless.multichannel_conflict_simulate(adj, k, float(p), rng=np.random)
        adj_list, adj_gk = wireless.multichannel_conflict_graph(graphs)
        assert len(adj_list) == k
        n = adj.shape[0]
        for c, a in enumerate(adj_list):
            a = sp.csr_matrix(a)
            assert np.array_equal(a.indptr, z["c%d|ch%d|indptr" % (ci, c)]) and np.array_equal(a.indices, z["c%d|ch%d|indices" % (ci, c)])
            assert np.array_equal(a.data, z["c%d|ch%d|data" % (ci, c)])
            assert (a != a.T).nnz == 0 and a.diagonal().sum() == 0
        assert adj_gk.shape == (k * n, k * n)
        assert np.array_equal(adj_gk.indptr, z["c%d|joint|indptr" % ci]) and np.array_equal(adj_gk.indices, z["c%d|joint|indices" % ci])
        assert np.array_equal(adj_gk.data, z["c%d|joint|data" % ci])
        # structure: a link's K copies form a clique, block k is channel k's graph
        dense = adj_gk.toarray()
        for k1 in range(k):
            assert np.array_equal(dense[k1 * n:(k1 + 1) * n, k1 * n:(k1 + 1) * n], adj_list[k1].toarray())
            for k2 in range(k):
                if k1 != k2:
                    assert np.array_equal(dense[k1 * n:(k1 + 1) * n, k2 * n:(k2 + 1) * n], np.eye(n))
    with pytest.raises(AssertionError):
        wireless.multichannel_conflict_graph([golden.scipy(0), golden.scipy(2)])


def test_compact_transfer_format_packs_what_the_ordinary_packer_packs():
    """dgcn_pack_compact_batch (include/dgcn.h: 16-bit local column ids, 16-bit degrees) on the host: expanding its output
    in NumPy gives exactly the block-diagonal CSR dgcn_pack_batch writes for the same graphs - entry order as given (an
    unsorted row stays unsorted), weights and max_degree included, an empty graph in the middle - in about half the bytes;
    the same structural validation (a column outside its graph is an error, not a wrapped 16-bit id)."""
    import ctypes as C
    from distgcn_amd import _lib, datagen
    from distgcn_amd.batch import _addresses, pack_csr_lists
    lib = _lib.load()
    hb = datagen.ba_test2_batch(60)
    ps, cs, ws = [], [], []
    for g, (n0, n1) in enumerate(hb.graph_slices()):
        e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
        ps.append((hb.row_ptr[n0:n1 + 1] - e0).astype(np.int64)); cs.append((hb.col_idx[e0:e1] - n0).astype(np.int64)); ws.append(hb.weights[n0:n1].copy())
    ps.insert(3, np.zeros(1, np.int64)); cs.insert(3, np.zeros(0, np.int64)); ws.insert(3, np.zeros(0))  # an empty graph
    cs[5][[0, 1]] = cs[5][[1, 0]]  # an unsorted row: kept as it is

    def compact(ps, cs, ws):
        B = len(ps)
        ap, cp, isz = _addresses(ps, 0)
        ai, _, _ = _addresses(cs, isz)
        aw, _, _ = _addresses(ws, 64)
        nn = (cp[:B] - 1).astype(np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        info, ci = _lib.DgcnPackInfo(), _lib.DgcnCompactInfo()
        _lib.check(lib.dgcn_pack_measure(vp(ap), vp(nn), B, isz, 1, C.byref(info), None), "measure")
        if lib.dgcn_pack_compact_layout(C.byref(info), C.byref(ci)) != 0:
            return 1, None, None, None
        buf = np.zeros(int(ci.total_bytes), np.uint8)
        rc = lib.dgcn_pack_compact_batch(vp(ap), vp(ai), vp(aw), vp(nn), B, isz, vp(buf), buf.nbytes, C.byref(info), C.byref(ci), 3)
        return rc, buf, info, ci

    rc, buf, info, ci = compact(ps, cs, ws)
    assert rc == 0 and int(ci.total_bytes) < 0.6 * int(info.total_bytes)
    B, n, e = int(info.num_graphs), int(info.num_nodes), int(info.num_edges)
    gp = buf[int(ci.off_graph_ptr):].view(np.int32)[:B + 1]
    ep = buf[int(ci.off_edge_ptr):].view(np.int32)[:B + 1]
    deg = buf[int(ci.off_deg):].view(np.uint16)[:n]
    col = buf[int(ci.off_col):].view(np.uint16)[:e]
    wt = buf[int(ci.off_weights):].view(np.float64)[:n]
    std, sinfo = pack_csr_lists(ps, cs, ws)
    ref = HostBatch.from_packed(std, sinfo)
    assert np.array_equal(gp, ref.graph_ptr) and np.array_equal(wt, ref.weights) and int(info.max_degree) == int(sinfo.max_degree)
    assert np.array_equal(ep, ref.row_ptr[ref.graph_ptr])
    # what k_expand_compact does: degrees -> row pointers, local ids + the graph's first vertex -> global ids
    row_ptr = np.concatenate([[0], np.cumsum(deg.astype(np.int64))])
    col_idx = col.astype(np.int64) + np.repeat(gp[:-1].astype(np.int64), np.diff(ep))
    assert np.array_equal(row_ptr, ref.row_ptr) and np.array_equal(col_idx, ref.col_idx)
    bad = [c.copy() for c in cs]
    bad[7][2] = ps[7].size + 70000  # outside the graph (and beyond 16 bits)
    assert compact(ps, bad, ws)[0] < 0 and b"outside" in lib.dgcn_last_error()
