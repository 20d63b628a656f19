#!/bin/bash
# round 3, call 17: C3 launch time by workgroup shape (what two co-resident workgroups buy over one)
for cfg in "" "DGCN_FUSED_BLOCK=1024" "DGCN_FUSED_LDS_PAD=40000" "DGCN_FUSED_PRIO=0" "DGCN_FUSED_PRIOG=0"; do
  echo "== $cfg"
  env $cfg python bench.py --cpu-seconds 0 --no-cpu-pool --no-spmm-probe --no-e2e --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
python tools/time_small.py 2>&1 | tail -8
DGCN_FUSED_BLOCK=1024 python tools/time_small.py 2>&1 | tail -8
