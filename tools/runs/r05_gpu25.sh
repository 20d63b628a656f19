#!/bin/bash
# round 5, call 25: k_big without the record stream (diag build: every walk re-reads its tile's first record group): the bound on what 16-bit records could buy
bash tools/build_diag.sh 2>&1 | grep -i error
for v in 0 1; do for c in er500 mc900; do echo -n "DGCN_BIG_DIAG_REC=$v "; DGCN_BIG_DIAG_REC=$v DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/run_general.py $c 200 20 256 2>/dev/null | grep big_solve; done; done
rm -f distgcn_amd/libdgcn_diag.so
