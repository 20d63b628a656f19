"""Child process of tests/test_gpu_wide.py / test_gpu_general.py: complete dit / cit / rollout searches and a plain solve on
three ragged ~900-vertex graphs (or ~`nodes`), results to an .npz: python _wide_witness.py out.npz [num_layer=1] [nodes=900].  Run twice - as built
and with a switch of the library set (DGCN_OPTIONS="wide1=0", applied by distgcn_amd/_lib.py through dgcn_set_option: one-layer
models layer by layer instead of csrc/wide.hip; "big_residual=0": the residual steps of deep models through the compaction launches + k_big + k_lgs
instead of one launch of k_big) - the two files must hold the same bytes / bits."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_path, num_layer=1, nodes=900):
    import torch
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import Engine, DeviceModel
    import scipy.sparse as sp
    rng = np.random.default_rng(99)
    mats, ws = [], []
    for n in (nodes, nodes - 29, nodes - 260 if nodes < 1200 else nodes - 400):
        ip, ix = datagen.er_graph(n, 0.012, rng)
        mats.append(sp.csr_matrix((np.ones(ix.size), ix, ip), shape=(n, n)))
        w = rng.random(n)
        w[rng.random(n) < 0.05] = 0.0  # zero weights inside live graphs
        ws.append(w)
    hb = HostBatch.from_scipy(mats, ws)
    layers = datagen.random_model(num_layer, 32, bias=True, last_act="leaky_relu", seed=12)
    eng = Engine("cuda:0")
    db = eng.upload(hb)
    dm = DeviceModel(layers, eng.device)
    out = {}
    r = eng.solve_fused(db, dm)
    eng.check_status(r["status"])
    out["plain_state"], out["plain_scores"] = r["state"].cpu().numpy(), r["scores"].cpu().numpy().ravel()
    out["plain_rounds"], out["plain_totals"] = r["rounds"].cpu().numpy(), r["totals"].cpu().numpy()
    for name, greedy, predict in (("dit", eng.GREEDY_ROUNDS, "mwis"), ("cit", eng.GREEDY_CENTRAL, "mwis"),
                                  ("rollout", eng.GREEDY_ROLLOUT, "mwis"), ("dit_mis", eng.GREEDY_ROUNDS, "mis")):
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
        res = eng.solve_residual(db, dm, state, predict=predict, greedy=greedy, max_rounds=1, beam=5,
                                 weight_features=predict != "mwis", want_scores=True)
        eng.check_status(res["status"])
        out[name + "_state"] = state.cpu().numpy()
        out[name + "_steps"] = np.array([res["steps"]])
        out[name + "_scores"] = res["scores"].cpu().numpy().ravel()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 900)
