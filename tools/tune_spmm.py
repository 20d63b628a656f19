#!/usr/bin/env python3
"""A/B of SpMM launch geometries in one process (interleaved rounds, per cdna guide rule 24)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd import _lib
from distgcn_amd.engine import Engine

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "er"
    hb = datagen.er_batch(500, 200, 0.1) if kind == "er" else datagen.ba_test2_batch(500)
    eng = Engine("cuda:0"); db = eng.upload(hb); lap = eng.supports(db)
    n = hb.num_nodes
    Z = torch.randn(n, 64, device="cuda"); out = torch.empty(n, 32, device="cuda")
    nnz_l = hb.num_edges + n
    bytes_ = nnz_l * 8 + (n + hb.num_graphs) * 4 + 3 * 4 * 32 * n
    variants = [dict(DGCN_SPMM_GLOBAL="1")]
    variants.append(dict(DGCN_SPMM_GLOBAL="1", DGCN_SPMM_SPLIT="1"))
    for split, block in itertools.product([1, 2, 4, 8], [256, 512, 1024]):
        variants.append(dict(DGCN_SPMM_SPLIT=str(split), DGCN_SPMM_BLOCK=str(block)))
    res = {i: [] for i in range(len(variants))}
    for rnd in range(5):
        for i, v in enumerate(variants):
            for k in ("DGCN_SPMM_GLOBAL", "DGCN_SPMM_ROWS", "DGCN_SPMM_BLOCK", "DGCN_SPMM_CSRCAP", "DGCN_SPMM_PAD", "DGCN_SPMM_SPLIT"):
                _lib.set_option(k[5:].lower(), {'DGCN_SPMM_CSRCAP': -1}.get(k, 0))
            [_lib.set_option(k[5:].lower(), int(val)) for k, val in v.items()]
            eng.timing(True)
            for _ in range(20):
                eng.spmm(lap, Z[:, 32:], 32, ldz=64, graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs,
                         max_nodes=hb.max_nodes, Y0=Z, ldy0=64, act="leaky_relu", out=out)
            torch.cuda.synchronize(); eng.timing(False)
            ms, cnt = eng.timing_read("spmm")
            res[i].append(ms / cnt * 1e3)
    for i, v in enumerate(variants):
        med = float(np.median(res[i]))
        print("%-55s median %7.2f us  min %7.2f us  %6.0f GB/s algorithmic" % (v, med, min(res[i]), bytes_ / med / 1e3))

if __name__ == "__main__":
    main()
