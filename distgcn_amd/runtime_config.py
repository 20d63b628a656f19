"""Flags of the reference (``runtime_config.py:6-34``) as a plain namespace.

The reference defines them through ``tf.compat.v1.flags`` and reads the global ``FLAGS`` inside
library code (``gcn/layers.py:175``, ``gcn/models.py:16``).  Same names and defaults here, no
TensorFlow/absl: ``FLAGS`` is a mutable namespace, ``flags.DEFINE_*`` adds fields (used by
``mwis_dqn_call.py:36-38`` style extensions), ``parse_argv`` accepts ``--name=value`` like absl.
"""
from __future__ import annotations

import sys
from types import SimpleNamespace

_DEFAULTS = dict(
    model="gcn_cheby", learning_rate=0.001, learning_decay=1.0, epochs=201, feature_size=32, hidden1=32,
    diver_num=32, dropout=0.0, weight_decay=5e-4, early_stopping=1000, max_degree=1, num_layer=20,
    backoff_prob=0.3, diver_out=32, timeout=300, datapath="./data/Random_Graph_Test", snr_db=10.0,
    training_set="IS4SAT", greedy=0, skip=False, wts_init="random", snapshot="", predict="mwis",
    epsilon=1.0, epsilon_min=0.001, epsilon_decay=0.985, gamma=1.0,
)


class _Flags(SimpleNamespace):
    def copy(self, **overrides):
        d = dict(self.__dict__)
        d.update(overrides)
        return _Flags(**d)


FLAGS = _Flags(**_DEFAULTS)


class _FlagsModule:
    """``flags.DEFINE_string('x', default, help)`` / ``flags.FLAGS`` as in absl."""
    FLAGS = FLAGS

    @staticmethod
    def _define(name, default, _help=""):
        if not hasattr(FLAGS, name):
            setattr(FLAGS, name, default)

    DEFINE_string = DEFINE_float = DEFINE_integer = DEFINE_bool = _define


flags = _FlagsModule()


def parse_argv(argv=None):
    """Apply ``--name=value`` / ``--name value`` arguments to FLAGS; returns the unparsed rest."""
    argv = list(sys.argv[1:] if argv is None else argv)
    rest = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--"):
            if "=" in a:
                name, val = a[2:].split("=", 1)
            elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
                name, val = a[2:], argv[i + 1]
                i += 1
            else:
                name, val = a[2:], "true"
            if hasattr(FLAGS, name):
                cur = getattr(FLAGS, name)
                if isinstance(cur, bool):
                    val = val.lower() in ("1", "true", "yes")
                elif isinstance(cur, int):
                    val = int(val)
                elif isinstance(cur, float):
                    val = float(val)
                setattr(FLAGS, name, val)
            else:
                rest.append(a)
        else:
            rest.append(a)
        i += 1
    return rest
