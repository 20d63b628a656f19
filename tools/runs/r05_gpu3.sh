#!/bin/bash
# round 5, call 3: MC900-l1 (the multi-channel launcher's own recipe) through wide.hip against the launch chain it replaces; test durations
python bench.py --config MC900-l1 --steps 1500 --cpu-seconds 10 --no-cpu-pool > gpurun_out/r05_bench_mc900_l1.json 2> gpurun_out/r05_bench_mc900_l1.err
tail -c 1500 gpurun_out/r05_bench_mc900_l1.json; echo
for w in 1 0; do for c in mc900 er500 er1500x0.01 er3000x0.003; do echo -n "DGCN_WIDE1=$w $c: "; DGCN_WIDE1=$w python tools/run_general.py $c 300 1 256 2>/dev/null | grep -v path | tr '\n' ';'; echo; done; done
python tools/run_iterative.py --help 2>&1 | head -5
timeout 900 python -m pytest tests/test_gpu_wide.py -q --tb=short -p no:cacheprovider --durations=12 2>&1 | tail -20
