#!/bin/bash
# round 4, call 46: searches with the tail on / off and the C5 bench lines on the final build (weight fetch by tile-owning waves only)
bash tools/runs/r04_gpu35.sh > /dev/null
O=gpurun_out/profiles_r04b
python bench.py --config C5 --cpu-seconds 25 2>/dev/null | tail -1 > $O/bench_c5.unprofiled.json
python bench.py --config C5 --graphs 256 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_c5_256.unprofiled.json
cut -c1-150 $O/iterative_tail_on_off.txt
python -c "
import json
for f in ('bench_c5','bench_c5_256'):
    d=json.load(open('$O/%s.unprofiled.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('tail',{}).get('ms_per_search'))
"
