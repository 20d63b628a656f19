#!/usr/bin/env python3
"""SpMM launch geometries on working sets LARGER than the 256 MiB Infinity Cache (HBM-bound measurement):
  one4000: one launch over a 4 000-graph ER batch (444 MB algorithmic per launch)
  rot8:    eight distinct 500-graph batches visited round-robin (55.5 MB per launch, 444 MB between re-uses)
Interleaved rounds in one process (cdna guide rule 24)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distgcn_amd import datagen
from distgcn_amd import _lib
from distgcn_amd.engine import Engine

KEYS = ("DGCN_SPMM_GLOBAL", "DGCN_SPMM_ROWS", "DGCN_SPMM_BLOCK", "DGCN_SPMM_CSRCAP", "DGCN_SPMM_PAD", "DGCN_SPMM_SPLIT")


def make(eng, hb):
    db = eng.upload(hb)
    return (hb, db, eng.supports(db), torch.randn(hb.num_nodes, 64, device="cuda"), torch.empty(hb.num_nodes, 32, device="cuda"))


def launch(eng, st):
    hb, db, lap, Z, out = st
    eng.spmm(lap, Z[:, 32:], 32, ldz=64, graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs, max_nodes=hb.max_nodes,
             Y0=Z, ldy0=64, act="leaky_relu", out=out)


def nbytes(hb, y0=True):
    return (hb.num_edges + hb.num_nodes) * 8 + (hb.num_nodes + hb.num_graphs) * 4 + (3 if y0 else 2) * 4 * 32 * hb.num_nodes


def main():
    eng = Engine("cuda:0")
    big = make(eng, datagen.er_batch(4000, 200, 0.1, first_index=50_000))
    rot = [make(eng, datagen.er_batch(500, 200, 0.1, first_index=i * 500)) for i in range(8)]
    variants = [dict()]
    for rows, block in (itertools.product([200, 100, 64], [256, 512, 1024]) if os.environ.get("DGCN_TUNE_FULL") else []):
        variants.append(dict(DGCN_SPMM_ROWS=str(rows), DGCN_SPMM_BLOCK=str(block)))
    variants.append(dict(DGCN_SPMM_GLOBAL="1"))
    variants.append(dict(DGCN_SPMM_CSRCAP="0"))
    for extra in sys.argv[1:]:  # "K=V,K=V"
        variants.append(dict(kv.split("=") for kv in extra.split(",")))
    res = {(w, i): [] for w in ("one4000", "rot8") for i in range(len(variants))}
    for rnd in range(3):
        for i, v in enumerate(variants):
            for k in KEYS:
                _lib.set_option(k[5:].lower(), {'DGCN_SPMM_CSRCAP': -1}.get(k, 0))
            [_lib.set_option(k[5:].lower(), int(val)) for k, val in v.items()]
            for w, sets, reps in (("one4000", [big], 6), ("rot8", rot, 3)):
                for st in sets:
                    launch(eng, st)
                torch.cuda.synchronize()
                eng.timing(True)
                for _ in range(reps):
                    for st in sets:
                        launch(eng, st)
                torch.cuda.synchronize(); eng.timing(False)
                ms, cnt = eng.timing_read("spmm")
                res[(w, i)].append(ms / cnt * 1e3)
    for w, hb in (("one4000", big[0]), ("rot8", rot[0][0])):
        for i, v in enumerate(variants):
            med = float(np.median(res[(w, i)]))
            print("%-8s %-50s median %8.2f us  %6.0f GB/s (+Y0: frac %.3f)  plain B_spmm frac %.3f" % (
                w, v, med, nbytes(hb) / med / 1e3, nbytes(hb) / med / 1e3 / 8000, nbytes(hb, False) / med / 1e3 / 8000))

if __name__ == "__main__":
    main()
