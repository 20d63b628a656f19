#!/bin/bash
# round 5, call 9: k_big2 with the search's columns in LDS; bench MC1500; the whole GPU suite
for c in mc1500 er1900x0.004; do python tools/run_general.py $c 100 20 256 2>/dev/null | grep -v path; done
python bench.py --config MC1500 --cpu-seconds 10 --no-cpu-pool > gpurun_out/r05_bench_mc1500.json 2> gpurun_out/r05_bench_mc1500.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_mc1500.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'], d['parity_full_size'], d['cpu_baseline']['value'])
PY
timeout 3000 python -m pytest tests -x -q -m gpu --tb=short -p no:cacheprovider --durations=15 > gpurun_out/r05_gpu9_suite.log 2>&1
tail -25 gpurun_out/r05_gpu9_suite.log
