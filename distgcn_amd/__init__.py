"""distgcn_amd - MI355X-native GCN-forward + local-greedy MWIS hot path of zhongyuanzhao/distgcn.

Drop-in modules (same names as the reference): ``heuristics``, ``gcn.models`` / ``gcn.layers`` /
``gcn.utils``, ``mwis_dqn_call``, ``mwis_gdpg_call``, ``runtime_config``, ``directory``.
Native boundary: ``libdgcn.so`` (``include/dgcn.h``), bound in ``_lib``; ``engine.Engine`` is the thin
device layer above it.  There is no CPU fallback: without the built library or without a GPU the
compute entry points raise ``DgcnError``.
"""
from ._lib import DgcnError  # noqa: F401

__version__ = "0.1.0"
