#!/bin/bash
# round 3, call 48: final gate - whole GPU suite, fuzz x 100, smoke, default bench line
python -m pytest tests -m gpu -q 2>&1 | tail -2
DGCN_FUZZ_CASES=100 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default:', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'], d['vs_baseline'], d['dtype'], d['scaling'])"
