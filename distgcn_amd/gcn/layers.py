"""``GraphConvolution`` (reference ``gcn/layers.py:149-216``) on the device: one layer
``out = act(sum_i S_i . (x . W_i) [+ b])`` for the supports ``[I, L]``."""
from __future__ import annotations

import numpy as np

from ..runtime_config import FLAGS


class GraphConvolution:
    def __init__(self, input_dim, output_dim, placeholders=None, dropout=0., sparse_inputs=False, act="relu",
                 bias=False, featureless=False, num_supports=None, seed=0, **kwargs):
        if featureless:
            raise NotImplementedError("featureless=True is never used on the reference's inference path")
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.act = act if isinstance(act, str) or act is None else getattr(act, "__name__", "linear")
        k = num_supports if num_supports is not None else 1 + int(FLAGS.max_degree)
        rng = np.random.default_rng(seed)
        lim = np.sqrt(6.0 / (self.input_dim + self.output_dim))
        self.vars = {}
        for i in range(k):
            w = (rng.uniform(-lim, lim, size=(self.input_dim, self.output_dim)).astype(np.float32)
                 if FLAGS.wts_init == "random" else np.zeros((self.input_dim, self.output_dim), np.float32))
            self.vars["weights_%d" % i] = w
        if bias:
            self.vars["bias"] = np.zeros(self.output_dim, np.float32)

    def layer_dict(self):
        ws = [self.vars["weights_%d" % i] for i in range(len([k for k in self.vars if k.startswith("weights_")]))]
        return {"weights": ws, "bias": self.vars.get("bias"), "act": self.act}

    def __call__(self, engine, device_batch, X=None):
        """Apply the layer to a batch; ``X`` is a device tensor ``[num_nodes, input_dim]`` or None for the
        constant row-normalised features."""
        from ..engine import DeviceModel
        return engine.forward(device_batch, DeviceModel([self.layer_dict()], engine.device), X=X)
