#!/bin/bash
# round 6, call 15: final state - the whole GPU suite, smoke(), the default bench line
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r06_bench_final.json 2>/dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_final.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print(d["roofline"]["bound"], d["roofline"]["paced_by"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
