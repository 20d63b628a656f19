#!/bin/bash
# round 5, call 7: phase clocks of k_big2 (diag build made on the box)
bash tools/build_diag.sh 2>&1 | grep -i error
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_big2.py mc1500 20 256
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_big2.py er1000x0.01 20 256
