#!/bin/bash
# round 4, call 20: C2 step time, three runs (is the dispatch inside dgcn_solve_batch visible at 16 us per step?)
mkdir -p gpurun_out
for i in 1 2 3; do python bench.py --config C2 --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'])"; done
python - <<'PY'
import time, torch, ctypes as C
from distgcn_amd import datagen, _lib
from distgcn_amd.engine import Engine, DeviceModel
eng = Engine("cuda:0"); hb = datagen.er_batch(500, 100, 0.1); db = eng.upload(hb); dm = DeviceModel(datagen.random_model(1, 32), "cuda:0")
out = eng.solve_buffers(db, False)
for _ in range(200): eng.solve_fused(db, dm, want_scores=False, out=out)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5000): eng.solve_fused(db, dm, want_scores=False, out=out)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("issue %.2f us per call, total %.2f us per call" % ((t1-t0)/5000*1e6, (t2-t0)/5000*1e6))
lib=eng.lib
t0=time.perf_counter()
for _ in range(20000): lib.dgcn_solve_path(C.byref(db.c), C.byref(dm.c))
print("dgcn_solve_path via ctypes %.2f us" % ((time.perf_counter()-t0)/20000*1e6))
PY
