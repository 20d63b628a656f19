#!/bin/bash
R=$PWD; O=$R/gpurun_out/profiles_r02; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/c5_iterative
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/c5_iterative" -- python3 $R/tools/run_iterative.py --graphs 64 --host 0 > "$O/c5_iterative.json" 2> "$O/c5_iterative.err"
head -5 $O/c5_iterative/*/*kernel_stats.csv; cat $O/c5_iterative.json | tail -3
