#!/bin/bash
# round 5, call 10: k_fused C3 with the second co-resident workgroup started late (DGCN_FUSED_STAGGER, units of 1 024 cycles)
for st in 0 2 4 6 8 12 16 24 0; do echo -n "stagger $st: "; DGCN_FUSED_STAGGER=$st python tools/run_fused.py er 300 20 500 2>/dev/null | grep -i "fused_solve\|us avg" | head -1; done
