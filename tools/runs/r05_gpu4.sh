#!/bin/bash
# round 5, call 4: HIP output against the restatement at the any-size path's own sizes; MC1500 before k_big2 (layer by layer)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q --tb=short -p no:cacheprovider --durations=10 2>&1 | tail -16
python bench.py --config MC1500 --steps 100 --cpu-seconds 10 --no-cpu-pool --no-spmm-probe --no-e2e > gpurun_out/r05_bench_mc1500_layered.json 2> gpurun_out/r05_bench_mc1500_layered.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_mc1500_layered.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernels'], d['parity_full_size'], d['cpu_baseline'])
PY
