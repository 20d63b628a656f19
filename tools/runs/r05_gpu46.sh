#!/bin/bash
# round 5, call 46: k_wide1's whole searches on ahead lists (two walks over state / flag bytes per round, two barriers): parity + times, off / on
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_general.py -m gpu -x -q 2>&1 | tail -3
for on in 0 1; do
  echo "DGCN_WIDE_AHEAD=$on"
  DGCN_WIDE_AHEAD=$on python bench.py --config MC900-l1 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MC900-l1', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
  DGCN_WIDE_AHEAD=$on python tools/run_general.py er3000x0.003 200 1 256 2>/dev/null | grep wide_solve
  DGCN_WIDE_AHEAD=$on python tools/run_general.py mc900 200 2 256 2>/dev/null | grep wide_solve
done
