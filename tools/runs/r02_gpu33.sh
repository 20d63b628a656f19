#!/bin/bash
python -m pytest tests/test_gpu_kernels.py -x -q -k "largest_first" 2>&1 | tail -2
R=$PWD; O=$R/gpurun_out/profiles_r02; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for n in bench_c4_l1:"--family ba --layers 1 --steps 500 --warmup 5 --cpu-seconds 0 --no-spmm-probe" bench_c4_l20:"--family ba --layers 20 --steps 300 --warmup 5 --cpu-seconds 0 --no-spmm-probe"; do
  name=${n%%:*}; args=${n#*:}
  rm -rf $O/$name
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 $R/bench.py $args > "$O/$name.json" 2> "$O/$name.err"
  head -4 $O/$name/*/*kernel_stats.csv
done
