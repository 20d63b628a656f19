#!/bin/bash
# round 6, call 22: tests of k_fused<.., GW>; the whole suite; C4 (4 000 BA graphs on one GPU) bench line
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "words_in_global" 2>&1 | tail -8 | tee gpurun_out/r06_gw_tests.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_gpu_suite_gw.txt
python bench.py --config C4 --cpu-seconds 6 --no-cpu-pool > gpurun_out/r06_bench_c4_full.json 2>/dev/null
DGCN_OPTIONS="fused_gw=0" python bench.py --config C4 --cpu-seconds 0 --no-cpu-pool > gpurun_out/r06_bench_c4_full_nogw.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_bench_c4_full", "r06_bench_c4_full_nogw"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("parity_full_size", {}).get("sets_differing"))
PY
