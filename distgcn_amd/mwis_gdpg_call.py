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
from .mwis_dqn_call import _State, solve_host_batch
from .runtime_config import FLAGS, flags  # noqa: F401


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
        hb = HostBatch.from_csr_lists([c.indptr.astype(np.int64) for c in csrs],
                                      [c.indices.astype(np.int64) for c in csrs],
                                      [np.asarray(w, dtype=np.float64).reshape(-1, self.feature_size)[:, 0] for w in wts_list])
        res = solve_host_batch(get_engine(), self.model, hb, self.flags.predict, mode, X=self._features(hb))
        out = []
        for g, (n0, n1) in enumerate(hb.graph_slices()):
            sel = np.flatnonzero(res["state"][n0:n1] == 1)
            out.append((set(int(i) for i in sel), np.float64(res["totals"][g])))
        return out


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
