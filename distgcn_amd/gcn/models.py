"""GCN_DQN / GCN2_DQN (reference ``gcn/models.py:441-577`` and ``:580-716``), inference only.

The reference builds a TF-1 graph and is queried with
``sess.run([model.outputs_softmax, model.pred], feed_dict)`` (``mwis_dqn_call.py:140-143``).  Here a
model is a list of layer parameters on the device; ``predict(state)`` returns the same pair
``(act_values float32 [N, out], action int64 [out])``, and ``forward_batch`` runs a whole block-diagonal
batch in one go.  Losses, optimisers and ``opt_op`` (training) are out of scope.

Variable names follow the reference's checkpoints (``<scope>/graphconvolution_{k}_vars/weights_{i}``,
``.../bias``; layer uid reset per model, ``models.py:538``) so ``load()`` reads the shipped ``model/``
directories and ``save()`` writes bundles TensorFlow's Saver could restore.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional

import numpy as np

from .. import checkpoint
from ..runtime_config import FLAGS as _GLOBAL_FLAGS

_VAR_RE = re.compile(r"^(?P<scope>.*?)/?graphconvolution_(?P<k>\d+)_vars/(?P<leaf>weights_(?P<i>\d+)|bias)$")


def layers_from_params(params: Dict[str, np.ndarray], model: str = "GCN_DQN", scope: Optional[str] = None,
                       act: str = "leaky_relu") -> List[dict]:
    """{checkpoint variable name: array} -> ordered layer dicts.

    GCN_DQN: hidden layers leaky_relu, last layer identity (``models.py:539-573``);
    GCN2_DQN: every layer, the last included, uses ``act`` (``models.py:673-708``).
    Optimiser slots (``.../Adam``, ``beta*_power``) are ignored.  With ``scope=None`` the scope that
    owns ``graphconvolution_1_vars/weights_0`` is used (shipped checkpoints: ``gcn_dqn``).
    """
    found: Dict[str, Dict[int, dict]] = {}
    for name, arr in params.items():
        m = _VAR_RE.match(name)
        if not m:
            continue
        lyr = found.setdefault(m.group("scope"), {}).setdefault(int(m.group("k")), {"weights": {}, "bias": None})
        if m.group("leaf") == "bias":
            lyr["bias"] = np.asarray(arr, dtype=np.float32)
        else:
            lyr["weights"][int(m.group("i"))] = np.asarray(arr, dtype=np.float32)
    if not found:
        raise KeyError("no graphconvolution_*_vars variables in the checkpoint")
    if scope is None:
        scope = sorted(found, key=lambda s: (s not in ("gcn_dqn", "gcn2_dqn", "model/gcn2_dqn"), len(s), s))[0]
    if scope not in found:
        raise KeyError("scope %r not in checkpoint (has %s)" % (scope, sorted(found)))
    by_k = found[scope]
    layers = []
    for k in range(1, len(by_k) + 1):
        if k not in by_k:
            raise KeyError("checkpoint lacks %s/graphconvolution_%d_vars" % (scope, k))
        ws = by_k[k]["weights"]
        layers.append({"weights": [ws[i] for i in range(len(ws))], "bias": by_k[k]["bias"]})
    for i, lyr in enumerate(layers):
        last = i == len(layers) - 1
        lyr["act"] = "linear" if (last and model == "GCN_DQN") else act
    return layers


def _glorot(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))  # gcn/inits.py glorot
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


class _GCNBase:
    """Shared machinery: parameters, checkpoint I/O, device residency, predict."""
    model_kind = "GCN_DQN"
    default_scope = "gcn_dqn"

    def _init_params(self, dims, num_supports, bias, wts_init, seed):
        rng = np.random.default_rng(seed)
        self.vars: Dict[str, np.ndarray] = {}
        for k in range(1, len(dims)):
            base = "%s/graphconvolution_%d_vars" % (self.scope, k)
            for i in range(num_supports):
                shape = (dims[k - 1], dims[k])
                if wts_init == "random":
                    w = _glorot(rng, shape)
                elif wts_init == "zeros":
                    w = np.zeros(shape, dtype=np.float32)
                else:
                    raise NameError("Unsupported wts_init: {}".format(wts_init))  # layers.py:183
                self.vars["%s/weights_%d" % (base, i)] = w
            if bias:
                self.vars[base + "/bias"] = np.zeros(dims[k], dtype=np.float32)
        self._device_model = None

    # ---- parameters -------------------------------------------------------------------------
    @property
    def layers(self) -> List[dict]:
        return layers_from_params(self.vars, self.model_kind, self.scope, self.act_name)

    def set_params(self, params: Dict[str, np.ndarray], strict_shapes: bool = True):
        """Assign variables by name (what Saver.restore does).  Accepts a foreign scope."""
        src = layers_from_params(params, self.model_kind, None, self.act_name)
        mine = self.layers
        if len(src) != len(mine):
            raise ValueError("checkpoint has %d layers, model has %d" % (len(src), len(mine)))
        for k, (s, m) in enumerate(zip(src, mine), start=1):
            base = "%s/graphconvolution_%d_vars" % (self.scope, k)
            if len(s["weights"]) != len(m["weights"]):
                raise ValueError("layer %d: checkpoint has %d supports, model has %d" % (k, len(s["weights"]), len(m["weights"])))
            for i, w in enumerate(s["weights"]):
                if strict_shapes and w.shape != m["weights"][i].shape:
                    raise ValueError("layer %d weights_%d: checkpoint shape %s, model shape %s"
                                     % (k, i, w.shape, m["weights"][i].shape))
                self.vars["%s/weights_%d" % (base, i)] = w.copy()
            if s["bias"] is not None:
                self.vars[base + "/bias"] = s["bias"].copy()
            elif base + "/bias" in self.vars:
                raise ValueError("layer %d: model has a bias, checkpoint does not" % k)
        if getattr(self, "skip", False):  # tf.layers.dense head of models.py:505-521
            for leaf in ("dense/kernel", "dense/bias"):
                hits = [k for k in params if k.endswith(leaf) and "Adam" not in k]
                if not hits:
                    raise ValueError("skip=True model: checkpoint lacks %s" % leaf)
                arr = np.asarray(params[hits[0]], dtype=np.float32)
                if arr.shape != self.vars[self.scope + "/" + leaf].shape:
                    raise ValueError("%s: checkpoint shape %s, model shape %s" % (leaf, arr.shape, self.vars[self.scope + "/" + leaf].shape))
                self.vars[self.scope + "/" + leaf] = arr.copy()
        self._device_model = None
        self._head_dev = None

    def load(self, name: str):
        """Restore from a model directory or bundle prefix (``mwis_dqn_call.py:188-192``).
        Unlike the reference a missing checkpoint is an error, not silently random weights."""
        self.set_params(checkpoint.load_bundle(name))
        return self

    def save(self, name: str):
        """Write ``<name>/model.ckpt`` as a TF V2 bundle (``mwis_dqn_call.py:194-195``)."""
        checkpoint.save_bundle(os.path.join(name, "model.ckpt"), self.vars)

    # ---- device -----------------------------------------------------------------------------
    def device_model(self, engine):
        from ..engine import DeviceModel
        if self._device_model is None or self._device_model.device != engine.device:
            self._device_model = DeviceModel(self.layers, engine.device)
        return self._device_model

    @property
    def has_head(self) -> bool:
        """True when ``outputs`` is not simply the last layer's activation (is_dual, models.py:651-653; skip, :505-521)."""
        return bool(self.is_dual or getattr(self, "skip", False))

    def forward_batch(self, engine, device_batch, X=None, x_const=None, mode: int = 0):
        """``model.outputs`` [num_nodes, out] for a whole batch, on the device (no host round trip)."""
        act = engine.forward(device_batch, self.device_model(engine), X=X, x_const=x_const, mode=mode)
        if self.is_dual:
            return engine.head_dual(device_batch, act)
        if getattr(self, "skip", False):
            import torch
            if getattr(self, "_head_dev", None) is None or self._head_dev[0].device != engine.device:
                self._head_dev = (torch.from_numpy(np.ascontiguousarray(self.vars[self.scope + "/dense/kernel"])).to(engine.device),
                                  torch.from_numpy(np.ascontiguousarray(self.vars[self.scope + "/dense/bias"])).to(engine.device))
            if x_const is None:
                x_const = float(np.float32(1.0 / self.input_dim))
            return engine.head_skip(act, self._head_dev[0], self._head_dev[1], X=X, x_const=x_const, in_dim=self.input_dim)
        return act

    def predict(self, state, engine=None):
        """``sess.run([outputs_softmax, pred])`` for ONE graph given the reference's ``state`` dict
        ({"features": COO tuple, "support": [COO tuples]}) -> (act_values [N, out] f32, action [out] i64)."""
        from ..api_common import get_engine, state_to_device
        engine = engine or get_engine()
        db, X, x_const = state_to_device(engine, state, self.input_dim)
        dm = self.device_model(engine)
        mode = 1 if engine.solve_supported(db, dm) else 0
        scores_d = self.forward_batch(engine, db, X=X, x_const=x_const, mode=mode)
        if scores_d.shape[1] == 1:
            action = engine.argmax(db, scores_d).cpu().numpy().astype(np.int64)  # models.py:526 pred
            return scores_d.cpu().numpy(), action
        scores = scores_d.cpu().numpy()
        return scores, np.argmax(scores, axis=0)


class GCN_DQN(_GCNBase):
    """``GCN_DQN(placeholders, input_dim, **kwargs)`` (``models.py:441-460``).

    ``placeholders`` (TF feed slots in the reference) is accepted for signature compatibility and
    ignored; sizes come from ``flags`` (default: the global FLAGS): ``num_layer``, ``hidden1``,
    ``diver_num``, ``max_degree``, ``wts_init`` - exactly the FLAGS the reference's ``_build`` reads.
    """
    model_kind = "GCN_DQN"
    default_scope = "gcn_dqn"

    def __init__(self, placeholders=None, input_dim=None, flags=None, name=None, seed=0, **kwargs):
        fl = flags or _GLOBAL_FLAGS
        self.scope = name or self.default_scope
        self.input_dim = int(input_dim if input_dim is not None else fl.feature_size)
        self.output_dim = int(fl.diver_num)
        self.num_layer = int(fl.num_layer)
        self.act_name = "leaky_relu"
        self.is_dual = False
        self.skip = bool(getattr(fl, "skip", False))
        hidden = int(fl.hidden1)
        dims = [self.input_dim] + [hidden] * (self.num_layer - 1) + [self.output_dim]
        self._init_params(dims, 1 + int(fl.max_degree), False, fl.wts_init, seed)
        if self.skip:
            # tf.compat.v1.layers.dense over concat([dense_input, activations[-1]]) (models.py:505-521): variables
            # <scope>/dense/kernel [F + D, D] and <scope>/dense/bias [D]; glorot-uniform kernel, or the reference's
            # +-identity pattern when wts_init == 'zeros' (:511-520)
            F, D = self.input_dim, self.output_dim
            if fl.wts_init == "zeros":
                k = np.zeros((F + D, D), dtype=np.float32)
                half = int(D / 2)
                k[0:half, list(range(0, D - 1, 2))] = -np.identity(half, dtype=np.float32)
                k[0:half, list(range(1, D, 2))] = np.identity(half, dtype=np.float32)
            else:
                k = _glorot(np.random.default_rng(seed + 1), (F + D, D))
            self.vars[self.scope + "/dense/kernel"] = k
            self.vars[self.scope + "/dense/bias"] = np.zeros(D, dtype=np.float32)


class GCN2_DQN(_GCNBase):
    """``GCN2_DQN(placeholders, hidden_dim, act, num_layer, bias, ...)`` (``models.py:580-613``):
    explicit hyper-parameters, optional bias, activation on every layer including the last."""
    model_kind = "GCN2_DQN"
    default_scope = "gcn2_dqn"

    def __init__(self, placeholders=None, hidden_dim=32, act="leaky_relu", num_layer=1, bias=False,
                 learning_rate=0.00001, learning_decay=1.0, weight_decay=5e-4, is_dual=False, is_noisy=False,
                 input_dim=None, output_dim=1, max_degree=None, wts_init=None, name=None, seed=0, **kwargs):
        fl = _GLOBAL_FLAGS
        self.scope = name or self.default_scope
        self.input_dim = int(input_dim if input_dim is not None else fl.feature_size)
        self.output_dim = int(output_dim)  # placeholders['labels'] is (None, 1) in mwis_gdpg_call.py:63
        self.hidden_dim = int(hidden_dim)
        self.num_layer = int(num_layer)
        self.act_name = act if isinstance(act, str) else getattr(act, "__name__", "leaky_relu")
        self.bias = bool(bias)
        self.is_dual = bool(is_dual)
        dims = [self.input_dim] + [self.hidden_dim] * (self.num_layer - 1) + [self.output_dim]
        md = int(max_degree if max_degree is not None else fl.max_degree)
        self._init_params(dims, 1 + md, self.bias, wts_init or fl.wts_init, seed)
