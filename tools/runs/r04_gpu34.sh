#!/bin/bash
# round 4, call 34: C5 / iterative-solver evidence with the tail kernel; full GPU suite on the build with tile_ops.h / tail.hip
O=gpurun_out/profiles_r04b; rm -rf $O; mkdir -p $O
python bench.py --config C5 --cpu-seconds 25 2>/dev/null | tail -1 > $O/bench_c5.unprofiled.json
python bench.py --config C5 --graphs 256 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_c5_256.unprofiled.json
python tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 20 2>/dev/null | grep "^{" > $O/iterative_mc900.txt
python tools/run_iterative.py --n 500 --p 0.1 --graphs 64 --layers 20 2>/dev/null | grep "^{" > $O/iterative_er500.txt
python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" > $O/iterative_c5.txt
python tools/tail_probe.py 2>&1 | grep -v amdgpu > $O/tail_probe.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_default.unprofiled.json
cat $O/iterative_c5.txt $O/iterative_mc900.txt; python -c "
import json
for f in ('bench_c5','bench_c5_256','bench_default'):
    d=json.load(open('$O/%s.unprofiled.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('tail'))
"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu34.log 2>&1; tail -4 gpurun_out/r04_gpu34.log
