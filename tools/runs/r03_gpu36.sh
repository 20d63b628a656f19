#!/bin/bash
# round 3, call 36: final check - whole GPU suite, smoke(), the default bench line and every --config line, un-profiled
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/r03_final_bench_default.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_final_bench_default.json').read()); print('default:', d['metric'], round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'], d.get('parity_full_size'))"
for c in C2 C4-share C5; do python bench.py --config $c --cpu-seconds 3 --parity-seconds 5 2>&1 | tail -1 > gpurun_out/r03_final_bench_$c.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_final_bench_$c.json').read()); print('$c:', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; done
python bench.py --config C4-share --layers 1 --cpu-seconds 3 --parity-seconds 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4-l1:', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
