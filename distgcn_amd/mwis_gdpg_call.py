"""Drop-in for the inference API of the reference's ``mwis_gdpg_call.py`` (``MWISSolver``, ``DQNAgent``).

``DQNAgent(input_flags).solve_mwis(adj_0, wts_0, train=False, grd=1.0) -> (set, total_wt)``
(``mwis_gdpg_call.py:200-235``), with ``GCN2_DQN(bias=True)`` as the model (``:678-688``).
Training (``replay``, target network) is out of scope.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import scipy.sparse as sp

from .api_common import as_csr, get_engine
from .batch import HostBatch
from .gcn import utils as gutils
from .gcn.models import GCN2_DQN
from .mwis_dqn_call import _State, solve_csr_lists, solve_host_batch
from .runtime_config import FLAGS, flags  # noqa: F401


# which -> (GCN re-run on every residual graph, completions ordered by GCN priority instead of weight)
ROLLOUT_VARIANTS = {"rollout": (True, False), "rollout00": (False, False), "rollout0": (False, True),
                    "rollout1": (True, True)}


class MWISSolver(object):
    def __init__(self, input_flags, memory_size=5000):
        self.flags = input_flags
        self.feature_size = int(input_flags.feature_size)
        self.delta = 0.000001
        self.gamma = input_flags.gamma
        self.epsilon = input_flags.epsilon
        self.epsilon_min = input_flags.epsilon_min
        self.epsilon_decay = input_flags.epsilon_decay
        self.learning_rate = input_flags.learning_rate
        self.model = None

    def _build_model(self):
        raise NotImplementedError

    def makestate(self, adj, wts_nn):
        """``mwis_gdpg_call.py:82-97``: ones (row-normalised) when predict == 'mwis', else w/(max+1e-9)."""
        wts_nn = np.asarray(wts_nn, dtype=np.float64).reshape(-1, self.feature_size)
        n = wts_nn.shape[0]
        if self.flags.predict == "mwis":
            features = np.ones([n, self.feature_size])
        else:
            features = np.multiply(np.ones([n, self.feature_size]), wts_nn / (np.amax(wts_nn) + 1e-9))
        raw = features.copy()
        lil = sp.lil_matrix(features)
        feats = gutils.preprocess_features(lil) if self.flags.predict == "mwis" else gutils.sparse_to_tuple(lil)
        return _State(features=feats, adj=as_csr(adj), max_degree=int(self.flags.max_degree), features_raw=raw)

    def predict(self, state):
        return self.model.predict(state, get_engine())

    def act(self, state, train=False):
        if train:
            raise NotImplementedError("train=True is outside the inference drop-in")
        return self.predict(state)

    def load(self, name):
        self.model.load(name)
        print("loaded " + name)

    def save(self, name):
        self.model.save(name)

    def _features(self, hb: HostBatch):
        """Device feature matrix for predict != 'mwis' (None means the constant 1/F rows)."""
        if self.flags.predict == "mwis":
            return None
        import torch
        x = np.empty((hb.num_nodes, self.feature_size), dtype=np.float32)
        for n0, n1 in hb.graph_slices():
            w = hb.weights[n0:n1]
            x[n0:n1, :] = (w / (np.amax(w) + 1e-9)).astype(np.float32)[:, None] if n1 > n0 else 0
        return torch.from_numpy(x).to(get_engine().device)

    def solve_mwis(self, adj_0, wts_0, train=False, grd=1.0):
        """GCN followed by LGS (``:200-235``) -> (mwis, total_wt)."""
        if train:
            raise NotImplementedError("train=True (replay memory) is outside the inference drop-in")
        return self.solve_mwis_batch([adj_0], [wts_0])[0]

    def solve_mwis_batch(self, adjs: Sequence, wts_list: Sequence, mode: str = "auto") -> List[tuple]:
        csrs = [as_csr(a) for a in adjs]
        ws = [np.asarray(w, dtype=np.float64).reshape(-1, self.feature_size)[:, 0] for w in wts_list]
        if self.flags.predict == "mwis":  # constant features: the lean path (one pack, one launch, cached buffers)
            res, gp = solve_csr_lists(get_engine(), self.model, [c.indptr for c in csrs], [c.indices for c in csrs], ws,
                                      self.flags.predict, mode)
        else:
            hb = HostBatch.from_csr_lists([c.indptr for c in csrs], [c.indices for c in csrs], ws)
            res, gp = solve_host_batch(get_engine(), self.model, hb, self.flags.predict, mode, X=self._features(hb)), hb.graph_ptr
        out = []
        for g in range(len(csrs)):
            sel = np.flatnonzero(res["state"][gp[g]:gp[g + 1]] == 1)
            out.append((set(sel.tolist()), np.float64(res["totals"][g])))
        return out


    # ---- iterative solvers on the same kernels ------------------------------
    # Every shape runs entirely on the device (solve_iterative_batch): graphs the fused kernel takes with the residual
    # graph as a mask applied while the LDS image is built, larger ones with the residual graph re-sliced by the
    # any-size path (csrc/general.hip).  The host loop below - SciPy re-slicing exactly as the reference does, with every
    # forward pass, greedy round and rollout completion on the device - remains for ``device_iterative = False``,
    # ``reference_ties`` and models with more than two supports.
    def _residual_scores(self, adj_nn, wts_nn):
        """(DeviceBatch, device scores [n,1]) of one residual graph: ``makestate`` + ``act``."""
        eng = get_engine()
        hb = HostBatch.from_csr_lists([adj_nn.indptr], [adj_nn.indices],
                                      [np.asarray(wts_nn, dtype=np.float64)[:, 0]])
        db = eng.upload(hb)
        dm = self.model.device_model(eng)
        mode = 1 if eng.solve_supported(db, dm) else 0
        scores = self.model.forward_batch(eng, db, X=self._features(hb), mode=mode)
        return eng, db, scores

    @staticmethod
    def _start(adj_0, wts_0, feature_size):
        adj_0 = as_csr(adj_0)
        wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], feature_size))
        return adj_0, wts, -np.ones(adj_0.shape[0])

    @staticmethod
    def _slice(adj_0, wts, nIS_vec):
        remain = nIS_vec == -1
        rmap = np.argwhere(remain)[:, 0]
        return as_csr(adj_0[remain, :][:, remain]), wts[remain, :], rmap

    def _gcn_wts(self, scores, wts_nn):
        act_vals = scores.cpu().numpy().flatten()
        return act_vals * wts_nn.flatten() if self.flags.predict == "mwis" else act_vals.astype(np.float64)

    # ---- device-resident variants: the residual graph is masked inside the fused kernel
    device_iterative = True  # False: always re-slice on the host (the path for shapes outside the fused kernel)

    def solve_iterative_batch(self, adjs: Sequence, wts_list: Sequence, which: str = "dit", b: int = 16):
        """``solve_mwis_dit`` / ``_cit`` / ``_rollout`` (and ``rollout00`` / ``rollout0`` / ``rollout1``) for
        many graphs at once, one step of the whole batch per call of ``dgcn_solve_residual_batch``
        (``Engine.solve_residual``); None when neither the fused kernel nor the any-size path takes the shapes."""
        import torch
        csrs = [as_csr(a) for a in adjs]
        hb = HostBatch.from_csr_lists([c.indptr for c in csrs],
                                      [c.indices for c in csrs],
                                      [np.asarray(w, dtype=np.float64).reshape(-1, self.feature_size)[:, 0] for w in wts_list])
        eng = get_engine()
        db = eng.upload(hb)
        dm = self.model.device_model(eng)
        path = eng.solve_path(db, dm) if hb.num_nodes else 0  # 1 fused kernel, 2 any-size device path (same results)
        if path == 0 or (which.startswith("rollout") and not 1 <= b <= 64):
            return None
        greedy = eng.GREEDY_ROLLOUT if which.startswith("rollout") else \
            {"dit": eng.GREEDY_ROUNDS, "cit": eng.GREEDY_CENTRAL}[which]
        rescore, by_prio = ROLLOUT_VARIANTS.get(which, (True, False))
        options, scores = (eng.COMPLETE_BY_PRIORITY if by_prio else 0), None
        if not rescore:  # one forward pass on the full graphs; every step re-uses its scores
            options |= eng.SCORES_GIVEN
            scores = self.model.forward_batch(eng, db, X=self._features(hb), mode=1 if eng.solve_supported(db, dm) else 0)
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
        res = eng.solve_residual(db, dm, state, predict=self.flags.predict, greedy=greedy, max_rounds=1, beam=b,
                                 weight_features=self.flags.predict != "mwis", options=options, scores=scores)
        eng.check_status(res["status"])
        st = res["state"].cpu().numpy()
        out = []
        for n0, n1 in hb.graph_slices():
            sel = np.flatnonzero(st[n0:n1] == 1)
            # np.dot(nIS_vec, wts) with undecided vertices (only left when no positive weight remains) at -1
            nis = np.where(st[n0:n1] == 1, 1.0, np.where(st[n0:n1] == 0, -1.0, 0.0))
            out.append((set(int(i) for i in sel), np.dot(nis, hb.weights[n0:n1].reshape(-1, 1))))
        return out

    def solve_mwis_dit(self, adj_0, wts_0, train=False, grd=1.0):
        """GCN embedded into the greedy iteration (``mwis_gdpg_call.py:278-318``): scores are
        recomputed on the residual graph before every round.  -> (mwis, best_IS_util)"""
        dev = self.solve_iterative_batch([adj_0], [wts_0], "dit") if self.device_iterative else None
        if dev is not None:
            return dev[0]
        adj_0, wts, nIS_vec = self._start(adj_0, wts_0, self.feature_size)
        best = np.array([0.0])
        while np.sum(nIS_vec == -1) > 0:
            adj_nn, wts_nn, rmap = self._slice(adj_0, wts, nIS_vec)
            if np.sum(wts_nn) <= 0:
                break
            eng, db, scores = self._residual_scores(adj_nn, wts_nn)
            res = eng.lgs(db, scores=scores, weights=db.weights if self.flags.predict == "mwis" else None,
                          max_rounds=1, want_totals=False)
            eng.check_status(res["status"])
            st = res["state"].cpu().numpy()
            nIS_vec[rmap[st == 1]] = 1
            nIS_vec[rmap[st == 2]] = 0
            best = np.dot(nIS_vec, wts)
        return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best

    def solve_mwis_cit(self, adj_0, wts_0, train=False, grd=1.0):
        """GCN + centralised argmax, one vertex per step (``mwis_gdpg_call.py:343-384``)."""
        dev = self.solve_iterative_batch([adj_0], [wts_0], "cit") if self.device_iterative else None
        if dev is not None:
            return dev[0]
        adj_0, wts, nIS_vec = self._start(adj_0, wts_0, self.feature_size)
        best = np.array([0.0])
        while np.sum(nIS_vec == -1) > 0:
            adj_nn, wts_nn, rmap = self._slice(adj_0, wts, nIS_vec)
            if np.sum(wts_nn) <= 0:
                break
            _, _, scores = self._residual_scores(adj_nn, wts_nn)
            pick = int(np.argmax(self._gcn_wts(scores, wts_nn)))
            nb_v = adj_nn.indices[adj_nn.indptr[pick]:adj_nn.indptr[pick + 1]]
            nIS_vec[rmap[pick]] = 1
            nIS_vec[rmap[nb_v]] = 0
            best = np.dot(nIS_vec, wts)
        return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best

    def solve_mwis_cgs_train(self, adj_0, wts_0, train=False, grd=1.0):
        """The ``CGCN-CGS`` scheduler of ``wireless_dqn_test.py:262-266`` (``mwis_gdpg_call.py:778-839``): with
        ``train=False`` the reference runs exactly the loop of ``solve_mwis_cit`` (its replay bookkeeping is all behind
        ``if train``), so inference is that solver; ``train=True`` memorises transitions for the optimiser - outside the
        inference drop-in."""
        if train:
            raise NotImplementedError("training is outside the inference drop-in")
        return self.solve_mwis_cit(adj_0, wts_0, train=False, grd=grd)

    def _rollout(self, which, adj_0, wts_0, b=16, rng=None, reference_ties=False):
        """The four rollout searches of the reference share one loop; ``ROLLOUT_VARIANTS[which]`` says
        whether the GCN is re-run on every residual graph and what orders the greedy completions.
        ``reference_ties=True`` (host-controlled path, needs ``rng`` with ``choice``, e.g. a seeded ``numpy.random``)
        breaks ties exactly as the reference does: candidate totals summed in ITS order - ``np.sum`` over the CPython
        set the completion was collected in - compared with ``==`` and drawn with one ``rng.choice`` per step, NumPy's
        default argsort for the candidates; with the same seed the reference's own result comes back."""
        import torch
        rescore, by_prio = ROLLOUT_VARIANTS[which]
        if reference_ties and rng is None:
            raise ValueError("reference_ties needs rng (the reference draws np.random.choice at every step)")
        if rng is None and self.device_iterative:
            dev = self.solve_iterative_batch([adj_0], [wts_0], which, b=b)
            if dev is not None:
                return dev[0]
        adj_0, wts, nIS_vec = self._start(adj_0, wts_0, self.feature_size)
        best = np.array([0.0])
        full_scores = None
        if not rescore:
            full_scores = self._residual_scores(adj_0, wts)[2].cpu().numpy()
        while np.sum(nIS_vec == -1) > 0:
            adj_nn, wts_nn, rmap = self._slice(adj_0, wts, nIS_vec)
            n = wts_nn.shape[0]
            if np.sum(wts_nn) <= 0:
                break
            eng = get_engine()
            if rescore:
                eng, db, scores = self._residual_scores(adj_nn, wts_nn)
                act_vals = scores.cpu().numpy().flatten()
            else:
                hb = HostBatch.from_csr_lists([adj_nn.indptr], [adj_nn.indices],
                                              [wts_nn[:, 0]])
                db = eng.upload(hb)
                act_vals = full_scores.flatten()[rmap]
            gcn_wts = act_vals * wts_nn.flatten() if self.flags.predict == "mwis" else act_vals.astype(np.float64)
            children = (np.argsort(-gcn_wts) if reference_ties else np.argsort(-gcn_wts, kind="stable"))[0:b]
            cand = wts_nn[children].copy()
            if len(cand) > 1:
                init = np.zeros((len(children), n), dtype=np.uint8)
                for i, child in enumerate(children):
                    init[i, child] = 3
                    init[i, adj_nn.indices[adj_nn.indptr[child]:adj_nn.indptr[child + 1]]] = 3
                prio = torch.from_numpy(np.ascontiguousarray(gcn_wts)).to(eng.device) if by_prio else db.weights
                ro = eng.lgs_masked(db, prio, torch.from_numpy(init).to(eng.device), len(children),
                                    sum_weights=db.weights)
                eng.check_status(ro["status"])
                if reference_ties:
                    # the reference's float: np.sum(wts_ro[list(ps)]) with ps the Python set greedy_search filled in
                    # pick order (descending completion priority), indices in the re-sliced (adj_ro) numbering
                    st = ro["state"].cpu().numpy()
                    order_key = gcn_wts if by_prio else wts_nn[:, 0]
                    for i in range(len(children)):
                        keep = init[i] == 0
                        renum = np.cumsum(keep) - 1
                        members = np.flatnonzero(st[i] == 1)
                        members = members[np.argsort(-order_key[members], kind="stable")]
                        ps = set()
                        for v in renum[members]:
                            ps.add(v)
                        wts_ro = wts_nn[keep]
                        cand[i, 0] += np.sum(wts_ro[list(ps)]) if (by_prio or not rescore) else np.sum(wts_ro[:, 0][list(ps)])
                else:
                    cand[:, 0] += ro["totals"].cpu().numpy()[:, 0]
            if reference_ties:
                ties = np.flatnonzero(cand == cand.max())
            else:
                # candidates that complete to the same set tie mathematically but not bit for bit (the sums run
                # in different orders): totals within 1e-12 relative count as tied
                ties = np.flatnonzero(np.isclose(cand, cand.max(), rtol=1e-12, atol=0.0))
            i_best = int(rng.choice(ties)) if rng is not None else int(ties[0])
            pick = int(children[i_best])
            nb_v = adj_nn.indices[adj_nn.indptr[pick]:adj_nn.indptr[pick + 1]]
            nIS_vec[rmap[pick]] = 1
            nIS_vec[rmap[nb_v]] = 0
            best = np.dot(nIS_vec, wts)
        return set(int(i) for i in np.argwhere(nIS_vec == 1).flatten()), best

    def solve_mwis_rollout(self, adj_0, wts_0, train=False, grd=1.0, b=16, rng=None, reference_ties=False):
        """Top-``b`` GCN candidates, each scored by its weight plus a greedy completion (by weight) of what is
        left; the GCN is re-run on every residual graph (``mwis_gdpg_call.py:596-659``).  On the device the
        whole step is one launch; on the host-re-slicing path the ``b`` completions are ONE launch of the
        masked greedy kernel.  The reference breaks score ties with ``np.random.choice`` (unseeded) and ranks
        with an unstable sort: here ties go to the first candidate / lower index unless ``rng`` (a
        ``numpy.random.Generator``) is given."""
        return self._rollout("rollout", adj_0, wts_0, b=b, rng=rng, reference_ties=reference_ties)

    def solve_mwis_rollout00(self, adj_0, wts_0, train=False, grd=1.0, b=16, rng=None, reference_ties=False):
        """``mwis_gdpg_call.py:413-472``: the GCN runs ONCE on the full graph; completions by weight."""
        return self._rollout("rollout00", adj_0, wts_0, b=b, rng=rng, reference_ties=reference_ties)

    def solve_mwis_rollout0(self, adj_0, wts_0, train=False, grd=1.0, b=16, rng=None, reference_ties=False):
        """``:474-533``: GCN once; completions ordered by the GCN priority, valued by weight."""
        return self._rollout("rollout0", adj_0, wts_0, b=b, rng=rng, reference_ties=reference_ties)

    def solve_mwis_rollout1(self, adj_0, wts_0, train=False, grd=1.0, b=16, rng=None, reference_ties=False):
        """``:535-594``: GCN on every residual graph; completions ordered by the GCN priority."""
        return self._rollout("rollout1", adj_0, wts_0, b=b, rng=rng, reference_ties=reference_ties)

    # ---- thin inference helpers of the reference class ------------------------------------------------
    @staticmethod
    def mellowmax(q_vec, omega, beta=None):
        """``mwis_gdpg_call.py:140-145``."""
        q_vec = np.asarray(q_vec)
        c = np.max(q_vec)
        return c + np.log(np.sum(np.exp(omega * (q_vec - c))) / np.size(q_vec)) / omega

    def utility(self, adj_0, wts_0, train=False):
        """``:147-160`` -> (gcn_wts = act_values, state).  (The reference passes ``act()``'s tuple on as if it
        were the array - SURVEY 2 notes the latent bug; this returns the array.)"""
        wts_nn = np.reshape(np.asarray(wts_0, dtype=np.float64), (-1, self.feature_size))
        state = self.makestate(as_csr(adj_0), wts_nn)
        return self.predict(state)[0], state

    def topology_encode(self, adj_0, wts_0, train=False):
        """``:189-198`` -> act_values [N, 1]."""
        return self.utility(adj_0, np.reshape(wts_0, (-1, 1)))[0]

    def schedule(self, adj_0, wts_0, train=False):
        """``:162-187`` -> (mwis, total_wt, state, act_vals): GCN followed by the local greedy search."""
        wts_nn = np.reshape(np.asarray(wts_0, dtype=np.float64), (-1, self.feature_size))
        act_vals, state = self.utility(adj_0, wts_nn)
        mwis, total = self.solve_mwis(adj_0, wts_nn)
        return mwis, total, state, act_vals

    def solve_mwis_util(self, adj_0, wts_0, wts_u, train=False, grd=1.0):
        """``:237-276`` (inference branch): the set is chosen with ``wts_0``, valued with ``wts_u``."""
        if train:
            raise NotImplementedError("train=True (replay memory) is outside the inference drop-in")
        mwis, _ = self.solve_mwis(adj_0, wts_0)
        wts_u = np.asarray(wts_u, dtype=np.float64)
        return mwis, np.sum(wts_u[sorted(mwis)])

    def _wrap(self, inner, adj_0, wts_0, **kw):
        """Per connected component (``mwis_gdpg_call.py:320-341, 386-411``; components via SciPy
        instead of NetworkX, vertices in ascending order)."""
        import scipy.sparse.csgraph as csg
        adj_0 = as_csr(adj_0)
        wts = np.reshape(np.asarray(wts_0, dtype=np.float64), (adj_0.shape[0], self.feature_size))
        ncomp, labels = csg.connected_components(adj_0, directed=False)
        total = np.array([0.0])
        chosen = set()
        which = {"solve_mwis_cit": "cit", "solve_mwis_rollout": "rollout"}.get(getattr(inner, "__name__", ""))
        if which and self.device_iterative and kw.get("rng") is None and not kw.get("reference_ties"):
            # all components advance together: they are the graphs of ONE device batch, one launch per solver step
            comps = [np.flatnonzero(labels == c) for c in range(ncomp)]
            dev = self.solve_iterative_batch([adj_0[c, :][:, c] for c in comps], [wts[c, :] for c in comps], which,
                                             b=kw.get("b", 16))
            if dev is not None:
                for comp, (sub, util) in zip(comps, dev):
                    total = total + util
                    chosen |= set(int(comp[i]) for i in sub)
                return chosen, total
        for c in range(ncomp):
            comp = np.flatnonzero(labels == c)
            sub, util = inner(adj_0[comp, :][:, comp], wts[comp, :], **kw)
            total = total + util
            chosen |= set(int(comp[i]) for i in sub)
        return chosen, total

    def solve_mwis_cit_wrap(self, adj_0, wts_0, train=False, grd=1.0):
        return self._wrap(self.solve_mwis_cit, adj_0, wts_0)

    def solve_mwis_rollout_wrap(self, adj_0, wts_0, train=False, grd=1.0, b=16, rng=None):
        return self._wrap(self.solve_mwis_rollout, adj_0, wts_0, b=b, rng=rng)


class DQNAgent(MWISSolver):
    def __init__(self, input_flags=None, memory_size=5000, seed=0):
        super(DQNAgent, self).__init__(input_flags or FLAGS, memory_size)
        self.model = self._build_model(seed)
        self.gamma = 1.0

    def _build_model(self, seed=0):
        return GCN2_DQN(None, hidden_dim=self.flags.hidden1, num_layer=self.flags.num_layer, bias=True,
                        learning_rate=self.flags.learning_rate, learning_decay=self.flags.learning_decay,
                        weight_decay=self.flags.weight_decay, input_dim=self.flags.feature_size,
                        max_degree=self.flags.max_degree, wts_init=self.flags.wts_init, seed=seed)

    def replay(self, batch_size):
        raise NotImplementedError("training is outside the inference drop-in")
