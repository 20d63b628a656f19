"""CPU: TF V2 bundle reader/writer and the model classes' checkpoint naming."""
import os

import numpy as np
import pytest

from distgcn_amd import checkpoint
from distgcn_amd.gcn.models import GCN_DQN, GCN2_DQN, layers_from_params
from distgcn_amd.runtime_config import FLAGS


def test_bundle_roundtrip(tmp_path, golden):
    m = "result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn"
    params = golden.params(m)
    params["scalar"] = np.float32(3.5)
    params["ints"] = np.arange(7, dtype=np.int64)
    checkpoint.save_bundle(str(tmp_path / "model.ckpt"), params)
    back = checkpoint.load_bundle(str(tmp_path))
    assert set(back) == set(params)
    for k in params:
        assert np.array_equal(back[k], params[k]) and back[k].shape == np.asarray(params[k]).shape


def test_corrupt_bundle_fails_crc(tmp_path, golden):
    checkpoint.save_bundle(str(tmp_path / "model.ckpt"), golden.params("result_DQNBA_deep_ld1_c32_l1_cheb1_diver1_mwis_dqn"))
    f = tmp_path / "model.ckpt.data-00000-of-00001"
    raw = bytearray(f.read_bytes())
    raw[0] ^= 0xFF
    f.write_bytes(bytes(raw))
    with pytest.raises(checkpoint.CheckpointError, match="crc32c"):
        checkpoint.load_bundle(str(tmp_path))


def test_missing_checkpoint_is_an_error(tmp_path):
    with pytest.raises(checkpoint.CheckpointError):
        checkpoint.load_bundle(str(tmp_path / "nope"))


def test_shipped_checkpoints_when_available(golden):
    root = "/root/reference/model"
    if not os.path.isdir(root):
        pytest.skip("reference tree not present (GPU box)")
    for m in golden.model_names:
        t = checkpoint.load_bundle(os.path.join(root, m))
        for k, v in golden.params(m).items():
            assert np.array_equal(t[k], v)


def test_model_classes_load_save(tmp_path, golden):
    fl = FLAGS.copy(feature_size=1, hidden1=32, num_layer=20, diver_num=1, max_degree=1)
    model = GCN_DQN(None, input_dim=1, flags=fl)
    assert len(model.layers) == 20 and model.layers[-1]["act"] == "linear" and model.layers[0]["act"] == "leaky_relu"
    params = golden.params("result_IS4SAT_deep_ld1_c32_l20_cheb1_diver1_mwis_dqn")
    model.set_params(params)
    model.save(str(tmp_path))
    again = GCN_DQN(None, input_dim=1, flags=fl).load(str(tmp_path))
    for a, b in zip(model.layers, again.layers):
        assert all(np.array_equal(x, y) for x, y in zip(a["weights"], b["weights"]))
    with pytest.raises(ValueError):
        GCN_DQN(None, input_dim=1, flags=fl.copy(num_layer=3)).set_params(params)
    m2 = GCN2_DQN(None, hidden_dim=32, num_layer=3, bias=True, input_dim=1)
    assert all(l["act"] == "leaky_relu" and l["bias"] is not None for l in m2.layers)
    assert sorted(m2.vars)[0].startswith("gcn2_dqn/graphconvolution_1_vars/")
    biased = layers_from_params(golden.params("result_DQNMED_deep_ld1_c16_l1_cheb1_diver1_mwis_dqn"))
    assert biased[0]["bias"] is not None
