"""CPU oracle for the GCN-forward + local-greedy MWIS hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference's algorithm for the path
named in BASELINE.json.  It exists to *check* the HIP implementation; it must never be
imported by the product package (``distgcn_amd``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

Pinning status
--------------
* ``normalize_adj`` / ``simple_polynomials`` / ``preprocess_features`` (A1-A3) and every
  ``local_greedy_search*`` / ``greedy_search`` variant (A8, A8', A9) are **pinned**: the
  reference's own functions (``gcn/utils.py``, ``heuristics.py``) were imported in the build
  container and their outputs stored in ``tests/golden/`` by ``oracle/make_golden.py``;
  ``tests/test_oracle_golden.py`` replays them.  ``greedy_utility`` stored in the reference's
  ``.mat`` files is a second, reference-authored pin for A9.
* The GCN forward (A4-A6: ``gcn/layers.py``, ``gcn/models.py``) executes inside TensorFlow
  in the reference.  TensorFlow is not installable here and the reference ships no stored
  activations, so that part is a restatement of ``gcn/layers.py:189-216`` and
  ``gcn/models.py:536-573 / 670-708`` that is **parity unpinned** at the TF boundary
  (anchored only by the closed form for l=1 and by end-to-end approximation ratios).

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

LEAKY_ALPHA = 0.2  # tf.nn.leaky_relu default; never overridden (gcn/models.py:553,562,583)


# =============================================================================
# A1-A3: graph preprocessing  (gcn/utils.py)
# =============================================================================
def normalize_adj(adj):
    """D^-1/2 A D^-1/2 with 0 for isolated vertices  (gcn/utils.py:120-127).

    The reference evaluates ``(A . Dinv)^T . Dinv`` in float64; the product order per entry is
    (A_vu * dinv[u]) * dinv[v], kept here so the float64 bits agree.
    """
    a = sp.coo_matrix(adj)
    deg = np.asarray(a.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    scale = sp.diags(dinv)
    return ((a @ scale).T @ scale).tocoo()


def simple_polynomials(adj, k):
    """[I, L, L^2, ...] with L = I - normalize_adj(adj) as COO triples  (gcn/utils.py:258-274)."""
    n = adj.shape[0]
    lap = sp.eye(n) - normalize_adj(adj)
    mats = [sp.eye(n), lap]
    while len(mats) < k + 1:
        mats.append(mats[-1] * lap)
    return [sparse_to_tuple(m) for m in mats]


def sparse_to_tuple(mx):
    """(coords[nnz,2], values[nnz], shape)  (gcn/utils.py:79-95)."""
    coo = mx if sp.isspmatrix_coo(mx) else mx.tocoo()
    coords = np.stack([coo.row, coo.col], axis=1)
    return coords, coo.data, coo.shape


def preprocess_features(features):
    """Row-normalise a sparse feature matrix, inf -> 0  (gcn/utils.py:98-106)."""
    rowsum = np.asarray(features.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        rinv = np.power(rowsum, -1.0)
    rinv[np.isinf(rinv)] = 0.0
    return sparse_to_tuple(sp.diags(rinv).dot(features))


def makestate(adj, wts_nn, feature_size, max_degree=1, flavour="dqn_call", predict="mwis"):
    """The three ``makestate`` flavours of the reference.

    * ``dqn_call``  - mwis_dqn_call.py:129-138  (ones * w/||w||, row-normalised)
    * ``dqn_test``  - mwis_dqn_test.py:162-169  (ones * w, row-normalised)
    * ``gdpg``      - mwis_gdpg_call.py:82-97   (plain ones when predict=='mwis', row-normalised;
                       otherwise ones * w/(max w + 1e-9), not normalised)
    """
    n = wts_nn.shape[0]
    w = np.reshape(wts_nn, (n, -1))
    ones = np.ones([n, feature_size])
    if flavour == "dqn_call":
        feats = preprocess_features(sp.lil_matrix(ones * (w / np.linalg.norm(w))))
    elif flavour == "dqn_test":
        feats = preprocess_features(sp.lil_matrix(ones * w))
    elif flavour == "gdpg":
        if predict == "mwis":
            feats = preprocess_features(sp.lil_matrix(ones))
        else:
            feats = sparse_to_tuple(sp.lil_matrix(ones * (w / (np.amax(w) + 1e-9))))
    else:
        raise ValueError(flavour)
    return {"features": feats, "support": simple_polynomials(adj, max_degree)}


# =============================================================================
# A4-A6: GCN forward (restatement of the TF graph; parity unpinned at the TF boundary)
# =============================================================================
def _tuple_to_csr(tup, dtype):
    coords, vals, shape = tup
    return sp.csr_matrix((np.asarray(vals).astype(dtype), (coords[:, 0], coords[:, 1])), shape=shape)


def _act(name, x):
    if name == "leaky_relu":
        return np.where(x > 0, x, x.dtype.type(LEAKY_ALPHA) * x)
    if name == "relu":
        return np.maximum(x, 0)
    if name in ("linear", "identity", None):
        return x
    raise ValueError(name)


def graph_convolution(x, supports, weights, bias=None, act="leaky_relu"):
    """One GraphConvolution layer  (gcn/layers.py:189-216).

    out = act( sum_i S_i . (x . W_i) [+ b] ): transform first, then aggregate (202-206);
    supports are summed in index order (208); dropout is the identity at inference.
    ``x`` is dense here; the reference's first layer holds it as a sparse tensor, which only
    changes the kernel TF picks, not the arithmetic.
    """
    out = None
    for s, w in zip(supports, weights):
        term = s @ (x @ w)
        out = term if out is None else out + term
    if bias is not None:
        out = out + bias
    return _act(act, out)


def gcn_layer_specs(params, model="GCN_DQN", scope="gcn_dqn", num_supports=2):
    """Turn a {variable name: array} dict (checkpoint naming, A11: mwis_dqn_call.py:188-192;
    layer uid reset at gcn/models.py:538) into an ordered list of layer dicts.

    GCN_DQN   (gcn/models.py:536-573): hidden layers leaky_relu, last layer identity.
    GCN2_DQN  (gcn/models.py:670-708): every layer, including the last, uses ``act``.
    """
    layers = []
    k = 1
    while True:
        base = "%s/graphconvolution_%d_vars" % (scope, k)
        if base + "/weights_0" not in params:
            break
        ws = [np.asarray(params["%s/weights_%d" % (base, i)]) for i in range(num_supports)]
        b = params.get(base + "/bias")
        layers.append({"weights": ws, "bias": None if b is None else np.asarray(b)})
        k += 1
    if not layers:
        raise KeyError("no graphconvolution variables under scope %r" % scope)
    for i, lyr in enumerate(layers):
        last = i == len(layers) - 1
        lyr["act"] = "linear" if (last and model == "GCN_DQN") else "leaky_relu"
    return layers


def gcn_forward(layers, state, dtype=np.float32, is_dual=False):
    """``sess.run([model.outputs_softmax, model.pred])``  (mwis_dqn_call.py:140-143).

    Returns (act_values[N, out] of ``dtype``, action = argmax over nodes, axis 0).
    TF casts the float64 feed to float32 (sparse placeholders are tf.float32), hence
    ``dtype=np.float32`` is the reference behaviour; ``np.float64`` gives the error yardstick.
    """
    supports = [_tuple_to_csr(t, dtype) for t in state["support"]]
    x = _tuple_to_csr(state["features"], dtype).toarray().astype(dtype)
    for lyr in layers:
        ws = [w.astype(dtype) for w in lyr["weights"]]
        b = None if lyr["bias"] is None else lyr["bias"].astype(dtype)
        x = graph_convolution(x, supports, ws, b, lyr["act"])
    if is_dual:  # gcn/models.py:651-653
        x = x[:, 0].mean(axis=0) + (x[:, 1:] - x[:, 1:].mean(axis=0))
    return x, np.argmax(x, axis=0)


def priority(act_vals, wts_nn, predict="mwis"):
    """``gcn_wts``  (mwis_dqn_call.py:230-235; mwis_gdpg_call.py:211-216): f32 * f64 -> f64."""
    if predict == "mwis":
        return np.multiply(np.asarray(act_vals).flatten(), np.asarray(wts_nn).flatten())
    return np.asarray(act_vals).flatten()


# =============================================================================
# A8 / A8': local greedy search and its instrumented twins  (heuristics.py:77-305)
# =============================================================================
def _lgs_core(adj, wts, nstep=None, want_overhead=False):
    """Shared body of the five reference variants.

    Per round, for every vertex still in ``remain`` (heuristics.py:90-114):
      nb = neighbours(v) & remain                              (94-95)
      nb empty                      -> v joins                 (96-98)
      w[v] > max w[nb]              -> v joins, nb excluded    (103-105)
      w[v] == max w[nb]             -> v joins iff v < the lowest-index
                                       neighbour holding that max      (106-111)
    then remain -= joined | excluded (114).  Decisions inside a round read only the
    round-start ``remain``.  Counters follow _stats (184-208) and _overhead (236-262).
    """
    w = np.array(wts).flatten()
    n = w.size
    mwis = set()
    nb_is = set()
    remain = set(range(n))
    rounds = p2p = bst = 0
    oh = np.zeros_like(w) if want_overhead else None
    budget = nstep
    while remain and (budget is None or budget):
        bst += len(remain)
        for v in remain:
            _, cols = np.nonzero(adj[v])
            nbrs = set(cols) & remain
            p2p += len(nbrs)
            if oh is not None:
                oh[v] += len(nbrs)
            if not nbrs:
                mwis.add(v)
                continue
            order = sorted(nbrs)
            w_nb = w[order]
            top = w_nb.max()
            wins = False
            if w[v] > top:
                wins = True
            elif w[v] == top:
                first = order[list(w_nb).index(w[v])]
                wins = v < first
            if wins:
                mwis.add(v)
                nb_is |= nbrs
                if oh is not None:
                    oh[v] += 1  # "mute signaling"
        remain = remain - mwis - nb_is
        rounds += 1
        if budget is not None:
            budget -= 1
    total = np.sum(w[list(mwis)])
    bst += len(mwis)
    return mwis, total, rounds, p2p, bst, oh, nb_is


def local_greedy_search(adj, wts):
    """heuristics.py:77-116 -> (set, total)."""
    r = _lgs_core(adj, wts)
    return r[0], r[1]


def local_greedy_search_count(adj, wts):
    """heuristics.py:119-160 -> (set, total, rounds)."""
    r = _lgs_core(adj, wts)
    return r[0], r[1], r[2]


def local_greedy_search_stats(adj, wts):
    """heuristics.py:163-209 -> (set, total, rounds, p2p, bst)."""
    r = _lgs_core(adj, wts)
    return r[0], r[1], r[2], r[3], r[4]


def local_greedy_search_overhead(adj, wts):
    """heuristics.py:212-263 -> (set, total, rounds, p2p, bst, overhead vector)."""
    r = _lgs_core(adj, wts, want_overhead=True)
    return r[0], r[1], r[2], r[3], r[4], r[5]


def local_greedy_search_nstep(adj, wts, nstep=1):
    """heuristics.py:266-305 -> (set, total, excluded-neighbour set) after at most nstep rounds."""
    r = _lgs_core(adj, wts, nstep=nstep)
    return r[0], r[1], r[6]


def lgs_vectorised(indptr, indices, wts, max_rounds=None):
    """Array form of the same rule, used to check big batches quickly.

    Key order (w desc, index asc): v joins iff it beats every residual neighbour.
    Returns (state[N] uint8: 1 joined / 2 excluded / 0 still remaining, rounds).
    """
    w = np.asarray(wts, dtype=np.float64).ravel()
    n = w.size
    indptr = np.asarray(indptr)
    indices = np.asarray(indices)
    rows = np.repeat(np.arange(n), np.diff(indptr))
    state = np.zeros(n, dtype=np.uint8)
    rounds = 0
    while (state == 0).any() and (max_rounds is None or rounds < max_rounds):
        live = (state[rows] == 0) & (state[indices] == 0)
        r, c = rows[live], indices[live]
        beaten = (w[c] > w[r]) | ((w[c] == w[r]) & (c < r))
        lose = np.zeros(n, dtype=bool)
        lose[r[beaten]] = True
        win = (state == 0) & ~lose
        excl = np.zeros(n, dtype=bool)
        excl[c[win[r]]] = True
        state[win] = 1
        state[excl & (state == 0)] = 2
        rounds += 1
    return state, rounds


# =============================================================================
# A9: centralised greedy  (heuristics.py:13-35)
# =============================================================================
def greedy_search(adj, wts):
    """Sort by weight descending, sweep, skip vertices adjacent to an earlier pick.

    The reference's ``np.argsort(-w)`` is an unstable sort: the order of equal weights is
    unspecified there.  This restatement uses a stable sort (ties by ascending index), which
    is the order under which the sweep equals ``local_greedy_search``.
    """
    w = np.array(wts).flatten()
    ranks = np.argsort(-w, kind="stable")
    mwis = set()
    blocked = set()
    for i in ranks:
        if i in blocked:
            continue
        _, cols = np.nonzero(adj[i])
        mwis.add(i)
        blocked |= set(cols)
    return mwis, np.sum(w[list(mwis)])


# =============================================================================
# A10: solve_mwis orchestration
# =============================================================================
def solve_mwis_gdpg(layers, adj, wts, feature_size=1, max_degree=1, predict="mwis", dtype=np.float32):
    """mwis_gdpg_call.py:200-235 -> (set, total_wt)."""
    adj = sp.csr_matrix(adj)
    wts_nn = np.reshape(wts, (np.asarray(wts).shape[0], feature_size))
    state = makestate(adj, wts_nn, feature_size, max_degree, "gdpg", predict)
    act_vals, _ = gcn_forward(layers, state, dtype)
    gcn_wts = priority(act_vals, wts_nn, predict)
    mwis, _ = local_greedy_search(adj, gcn_wts)
    return mwis, np.sum(wts_nn[list(mwis), 0])


def solve_mwis_dqn(layers, adj, wts, feature_size=1, max_degree=1, predict="mwis", dtype=np.float32):
    """mwis_dqn_call.py:198-261 (inference branch) -> (set of original ids, total_wt, 1.0).

    Zero-weight vertices are deleted first and indices mapped back through ``kp_nodes``
    (202-207, 241).
    """
    adj = sp.csr_matrix(adj)
    w0 = np.asarray(wts).flatten()
    keep = np.where(w0 > 0)[0]
    sub = adj[keep][:, keep]
    wts_nn = w0[keep].reshape(len(keep), 1)
    state = makestate(sub, wts_nn, feature_size, max_degree, "dqn_call", predict)
    act_vals, _ = gcn_forward(layers, state, dtype)
    gcn_wts = priority(act_vals, wts_nn, predict)
    mwis, _ = local_greedy_search(sub, gcn_wts)
    solu = list(mwis)
    return set(keep[solu]), np.sum(wts_nn[solu, 0]), 1.0


# =============================================================================
# iterative solvers built on the same forward + greedy pieces
# (mwis_gdpg_call.py:278-411, 596-659).  ``scores_fn(adj_nn, wts_nn) -> act_vals [n, 1]`` stands for
# ``makestate`` + ``act`` (so tests can plug either this module's forward or the C twin's).
# =============================================================================
def _default_scores_fn(layers, feature_size=1, max_degree=1, predict="mwis", dtype=np.float32):
    def fn(adj_nn, wts_nn):
        state = makestate(sp.csr_matrix(adj_nn), wts_nn, feature_size, max_degree, "gdpg", predict)
        return gcn_forward(layers, state, dtype)[0]
    return fn


def _residual(adj_0, wts, nIS_vec):
    remain = nIS_vec == -1
    rmap = np.argwhere(remain)[:, 0]
    adj_nn = adj_0[remain, :][:, remain]
    return adj_nn, wts[remain, :], rmap


def solve_mwis_dit(scores_fn, adj_0, wts_0, predict="mwis"):
    """GCN re-run on the residual graph before every greedy round (mwis_gdpg_call.py:278-318)."""
    adj_0 = sp.csr_matrix(adj_0)
    wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], -1))
    nIS_vec = -np.ones(adj_0.shape[0])
    best = np.array([0.0])
    while np.sum(nIS_vec == -1) > 0:
        adj_nn, wts_nn, rmap = _residual(adj_0, wts, nIS_vec)
        if np.sum(wts_nn) <= 0:
            break
        act_vals = scores_fn(adj_nn, wts_nn)
        gcn_wts = priority(act_vals, wts_nn, predict)
        sol, _, nb = local_greedy_search_nstep(adj_nn, gcn_wts, nstep=1)
        nIS_vec[rmap[list(sol)]] = 1
        nIS_vec[rmap[list(nb)]] = 0
        best = np.dot(nIS_vec, wts)
    return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best


def solve_mwis_cit(scores_fn, adj_0, wts_0, predict="mwis"):
    """GCN + centralised argmax, one vertex per step (mwis_gdpg_call.py:343-384)."""
    adj_0 = sp.csr_matrix(adj_0)
    wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], -1))
    nIS_vec = -np.ones(adj_0.shape[0])
    best = np.array([0.0])
    while np.sum(nIS_vec == -1) > 0:
        adj_nn, wts_nn, rmap = _residual(adj_0, wts, nIS_vec)
        if np.sum(wts_nn) <= 0:
            break
        gcn_wts = priority(scores_fn(adj_nn, wts_nn), wts_nn, predict)
        pick = int(np.argmax(gcn_wts))
        _, nb_v = np.nonzero(adj_nn[pick])
        nIS_vec[rmap[pick]] = 1
        nIS_vec[rmap[nb_v]] = 0
        best = np.dot(nIS_vec, wts)
    return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best


def solve_mwis_cgs_train(scores_fn, adj_0, wts_0, predict="mwis"):
    """``solve_mwis_cgs_train(train=False)`` (mwis_gdpg_call.py:778-839): line for line the loop of solve_mwis_cit
    (:343-384) - the ``buffers`` / ``memorize`` bookkeeping only runs ``if train`` - so the same restatement serves."""
    return solve_mwis_cit(scores_fn, adj_0, wts_0, predict)


def solve_mwis_rollout(scores_fn, adj_0, wts_0, b=16, predict="mwis", rng=None, rescore=True, by_priority=False,
                       reference_ties=False):
    """Top-b GCN candidates, each scored by its weight plus a greedy completion of the residual
    (mwis_gdpg_call.py:596-659).  The reference breaks score ties with ``np.random.choice`` and ranks
    with an unstable sort; here ties go to the first candidate / lower index unless ``rng`` is given.
    Variants (mwis_gdpg_call.py:413-594): ``rescore=False`` runs the GCN once on the full graph (rollout00,
    rollout0); ``by_priority=True`` orders the greedy completions by the GCN priority instead of the weight,
    still valuing them by weight (rollout0, rollout1).
    ``reference_ties=True`` reproduces the reference's own tie handling bit for bit - candidate totals summed in
    its order (``np.sum(wts_ro[list(ps)])`` over the CPython set), exact ``scores == scores.max()``, one
    ``rng.choice`` per step (pass ``numpy.random`` seeded like the run to replay) and NumPy's default argsort: the
    mode tests/test_oracle_golden.py uses against the reference-executed vectors (tests/golden/ref_exec.npz)."""
    adj_0 = sp.csr_matrix(adj_0)
    wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], -1))
    nIS_vec = -np.ones(adj_0.shape[0])
    best = np.array([0.0])
    full_scores = None if rescore else np.asarray(scores_fn(adj_0, wts))
    while np.sum(nIS_vec == -1) > 0:
        adj_nn, wts_nn, rmap = _residual(adj_0, wts, nIS_vec)
        n = wts_nn.shape[0]
        if np.sum(wts_nn) <= 0:
            break
        act_vals = scores_fn(adj_nn, wts_nn) if rescore else full_scores[rmap]
        gcn_wts = priority(act_vals, wts_nn, predict)
        children = (np.argsort(-gcn_wts.flatten()) if reference_ties else np.argsort(-gcn_wts.flatten(), kind="stable"))[0:b]
        scores = wts_nn[children].copy()
        if len(scores) > 1:
            for i, child in enumerate(children):
                keep = np.ones((n,), dtype=bool)
                keep[child] = False
                _, nb_v = np.nonzero(adj_nn[child])
                keep[nb_v] = False
                adj_ro = adj_nn[keep, :][:, keep]
                if by_priority or (reference_ties and not rescore):
                    # rollout00 / 0 / 1 (mwis_gdpg_call.py:456-458, 517-519, 578-580) re-sum the completion themselves
                    ps, _ = greedy_search(adj_ro, gcn_wts.flatten()[keep] if by_priority else wts_nn[keep])
                    ss = np.sum(wts_nn[keep][list(ps)]) if reference_ties else np.sum(wts_nn[keep][sorted(ps)])
                else:
                    _, ss = greedy_search(adj_ro, wts_nn[keep])
                scores[i] += ss
        if reference_ties:
            ties = np.flatnonzero(scores == scores.max())
        else:
            # candidates that complete to the same set tie mathematically but not bit for bit (the sums run in
            # different orders): totals within 1e-12 relative count as tied
            ties = np.flatnonzero(np.isclose(scores, scores.max(), rtol=1e-12, atol=0.0))
        i_best = int(rng.choice(ties)) if rng is not None else int(ties[0])
        pick = int(children[i_best])
        _, nb_v = np.nonzero(adj_nn[pick])
        nIS_vec[rmap[pick]] = 1
        nIS_vec[rmap[nb_v]] = 0
        best = np.dot(nIS_vec, wts)
    return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best


def _components(adj):
    import scipy.sparse.csgraph as csg
    ncomp, labels = csg.connected_components(sp.csr_matrix(adj), directed=False)
    return [np.flatnonzero(labels == c) for c in range(ncomp)]


def solve_wrap(inner, scores_fn, adj_0, wts_0, reference_mapping=False, **kw):
    """``solve_mwis_cit_wrap`` / ``solve_mwis_rollout_wrap`` (mwis_gdpg_call.py:320-341, 386-411): run
    ``inner`` per connected component and add the utilities.

    ``reference_mapping=True`` reproduces the reference to the letter, including a latent bug: it takes the
    components from NetworkX as Python sets, slices the adjacency with a boolean mask (ascending vertex order) but maps
    the component-local solution back through ``list(component_set)[i]`` - CPython's set iteration order, which is not
    ascending for every component (fixture g01: {93, 21} iterates as [93, 21]).  The returned set can then hold a vertex
    the component's solver did not pick, and its weight no longer equals the returned total (26.607 vs 27.070 on g01).
    The default maps through the ascending order the slice used; totals are identical either way."""
    adj_0 = sp.csr_matrix(adj_0)
    wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], -1))
    total = np.array([0.0])
    chosen = set()
    if reference_mapping:
        import networkx as nx
        graph = nx.from_scipy_sparse_array(adj_0) if hasattr(nx, "from_scipy_sparse_array") else nx.from_scipy_sparse_matrix(adj_0)
        for comp_set in nx.connected_components(graph):
            order = list(comp_set)
            mask = np.zeros(adj_0.shape[0], dtype=bool)
            mask[order] = True
            sub, util = inner(scores_fn, adj_0[mask, :][:, mask], wts[mask, :], **kw)
            total = total + util
            chosen |= set(int(order[i]) for i in sub)
        return chosen, total
    for comp in _components(adj_0):
        sub, util = inner(scores_fn, adj_0[comp, :][:, comp], wts[comp, :], **kw)
        total = total + util
        chosen |= set(int(comp[i]) for i in sub)
    return chosen, total


def margin_risk(indptr, indices, prio, state, delta, wabs=None):
    """SURVEY 7.3(c), checker for dgcn_margin_risk_batch: the number of excluded vertices (state 2) without a member
    neighbour u (state 1) whose priority leads by more than delta * (|w_u| + |w_v|).  The local greedy search
    (heuristics.py:77-116) returns the unique independent set in which every excluded vertex has a member neighbour
    ahead of it in (priority desc, index asc); zero risky vertices means no score error up to ``delta`` changes it."""
    prio = np.asarray(prio, dtype=np.float64)
    wabs = np.ones_like(prio) if wabs is None else np.abs(np.asarray(wabs, dtype=np.float64))
    state = np.asarray(state)
    risky = 0
    for v in np.flatnonzero(state == 2):
        nb = indices[indptr[v]:indptr[v + 1]]
        nb = nb[state[nb] == 1]
        if not np.any(prio[nb] - prio[v] > delta * (wabs[nb] + wabs[v])):
            risky += 1
    return risky
