#!/bin/bash
# round 4, call 33: k_big with the first record group of an aggregation requested before the preceding transform
for c in ER500 MC900; do
  python bench.py --config $c --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
python bench.py --any-size-path --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 any-size', d['ms_per_step'], d['roofline']['avg_launch_us'])"
timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider 2>&1 | tail -2
