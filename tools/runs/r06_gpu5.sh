#!/bin/bash
# round 6, call 5: the residual launch's automatic cluster form (C5), its fault recovery, C5 at 64 / 256 / 512 graphs
timeout 900 python -m pytest tests/test_gpu_api.py -x -q -p no:cacheprovider -k "cluster or c5 or bench_contract" 2>&1 | tail -5 | tee gpurun_out/r06_gpu5_tests.txt
for c in 0 -1; do echo "fused_cluster=$c:"; DGCN_OPTIONS="fused_cluster=$c" python tools/run_iterative.py 2>&1 | grep -v amdgpu | tail -8; done | tee gpurun_out/r06_gpu5_iterative.txt
python bench.py --config C5 > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err; tail -c 600 gpurun_out/r06_bench_c5.json | head -c 300; echo
DGCN_OPTIONS="fused_cluster=0" python bench.py --config C5 --no-cpu-pool --cpu-seconds 0 > gpurun_out/r06_bench_c5_nocluster.json 2>/dev/null
python bench.py --config C5 --graphs 256 --cpu-seconds 0 --no-cpu-pool > gpurun_out/r06_bench_c5_256.json 2>/dev/null
python bench.py --config C5 --graphs 512 --cpu-seconds 0 --no-cpu-pool > gpurun_out/r06_bench_c5_512.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_bench_c5", "r06_bench_c5_nocluster", "r06_bench_c5_256", "r06_bench_c5_512"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d.get("without_launch_events"))
    except Exception as e:
        print(f, "failed", e)
PY
