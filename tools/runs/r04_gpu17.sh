#!/bin/bash
# round 4, call 17: dgcn_solve_batch's whole path in ONE launch for graphs beyond the fused kernel (k_big: supports while the
# records are written, every layer, priority, greedy search): tests, then the two bench shapes
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_api.py -m gpu -q --tb=short -p no:cacheprovider -x -k "general or big or large or host_solver or native" > gpurun_out/r04_gpu17.log 2>&1
tail -3 gpurun_out/r04_gpu17.log
for c in ER500 MC900; do python bench.py --config $c --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 10 --steps 300 2>/dev/null | tail -1 > gpurun_out/r04_onelaunch_$c.json; done
python - <<'PY'
import json
for c in ("ER500","MC900"):
    d=json.load(open("gpurun_out/r04_onelaunch_%s.json"%c)); print(c, d["value"], d["ms_per_step"], d["kernels"], d["parity_full_size"]["sets_differing"], d["parity_full_size"]["max_err"])
PY
