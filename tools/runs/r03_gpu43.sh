#!/bin/bash
# round 3, call 43: issue-priority settings of k_fused once more, after the block-major records (tools/ab_fused.py, interleaved)
python tools/ab_fused.py "" "DGCN_FUSED_PRIO=0" "DGCN_FUSED_PRIO=3" "DGCN_FUSED_PRIO=8" "DGCN_FUSED_PRIOG=0" "DGCN_FUSED_PRIOG=2" "DGCN_FUSED_PRIOG=3" "DGCN_FUSED_PRIO=3,DGCN_FUSED_PRIOG=2" 2>&1 | grep -v amdgpu.ids | tail -10
