#!/bin/bash
# round 4, call 48: final check of the round's last build: smoke(), the whole GPU suite, the default bench line
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu48.log 2>&1; tail -3 gpurun_out/r04_gpu48.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_gpu48_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r04_gpu48_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
