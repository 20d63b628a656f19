#!/usr/bin/env python3
"""A/B of whole library builds (tools/placement_sweep.sh): a fresh child process per (library, round), rounds interleaved over the
libraries; every child times the C3 launch (500 ER(200, 0.1) graphs, 20 layers; DGCN_AB_GRAPHS / DGCN_AB_KIND as in ab_fused.py)
with the library's own event pairs and prints a CRC of scores and states, which must agree across the libraries.
   python tools/ab_libs.py build/sweep/libdgcn_a.so build/sweep/libdgcn_b.so ...      (DGCN_AB_ROUNDS, default 3)"""
import os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("DGCN_AB_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from distgcn_amd import datagen
    from distgcn_amd.engine import Engine, DeviceModel
    nB = int(os.environ.get("DGCN_AB_GRAPHS", "500"))
    hb = datagen.ba_test2_batch(nB) if os.environ.get("DGCN_AB_KIND", "er") == "ba" else datagen.er_batch(nB, 200, 0.1)
    eng = Engine("cuda:0"); db = eng.upload(hb); model = DeviceModel(datagen.random_model(20, 32), "cuda:0")
    out = eng.solve_buffers(db, True)
    for _ in range(400):
        eng.solve_fused(db, model, out=out)
    torch.cuda.synchronize()
    meds = []
    for _ in range(5):
        eng.timing(True)
        for _ in range(100):
            eng.solve_fused(db, model, out=out)
        torch.cuda.synchronize(); eng.timing(False)
        ms, n = eng.timing_read("fused_solve")
        meds.append(ms / n * 1e3)
    crc = zlib.crc32(out["state"].cpu().numpy().tobytes(), zlib.crc32(out["scores"].cpu().numpy().tobytes()))
    print("RESULT %.2f %.2f %08x" % (float(np.median(meds)), min(meds), crc))
    sys.exit(0)

libs = sys.argv[1:]
rounds = int(os.environ.get("DGCN_AB_ROUNDS", "3"))
res = {l: [] for l in libs}
crcs = {}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, DGCN_AB_CHILD="1", DGCN_LIB=os.path.abspath(l))
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("RESULT")]
        if not line:
            print("%s: FAILED\n%s" % (l, p.stderr[-2000:]), flush=True)
            continue
        med, mn, crc = line[0].split()[1:]
        res[l].append(float(med)); crcs.setdefault(l, set()).add(crc)
        print("round %d %-40s median %7.2f min %7.2f crc %s" % (r, os.path.basename(l), float(med), float(mn), crc), flush=True)
print()
allc = set().union(*crcs.values()) if crcs else set()
for l in libs:
    if res[l]:
        print("%-40s %s  -> best-of-rounds %7.2f us, mean %7.2f" % (os.path.basename(l), " ".join("%7.2f" % x for x in res[l]), min(res[l]), sum(res[l]) / len(res[l])))
print("results identical across libraries: %s" % (len(allc) == 1))
