#!/bin/bash
# round 5, call 47: the rounds on ahead lists in k_lgs (whole searches without statistics) and k_big2 too: the whole GPU suite, then times off / on
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for on in 0 1; do
  echo "DGCN_WIDE_AHEAD=$on"
  DGCN_WIDE_AHEAD=$on python tools/run_lgs.py 2>/dev/null | tail -3
  DGCN_WIDE_AHEAD=$on python bench.py --config MC1500 --steps 200 --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MC1500', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
