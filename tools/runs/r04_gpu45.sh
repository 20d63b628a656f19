#!/bin/bash
# round 4, call 45: k_big too fetches weight fragments only on waves that own a tile; searches with the tail on / off; bench lines
bash tools/runs/r04_gpu35.sh
for c in ER500 MC900; do
  python bench.py --config $c --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_fuzz.py tests/test_gpu_tail.py -x -q -p no:cacheprovider 2>&1 | tail -2
