#!/bin/bash
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_final.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['e2e']['value'], d['single_graph'])"
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single -- python3 $R/tools/time_small.py 1 > $R/gpurun_out/prof_single.log 2>&1
cd $R
f=$(ls gpurun_out/prof_single/*/*kernel_stats.csv | head -1); head -5 $f; cp $f gpurun_out/r02_single_graph_kernel_stats.csv
cat gpurun_out/prof_single.log | tail -2
