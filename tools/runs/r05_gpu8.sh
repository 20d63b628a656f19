#!/bin/bash
# round 5, call 8: k_big2 with Z0 and the lo epilogue in S3, laundered per-tile addresses: timing, phase clocks, tests
for c in mc1500 er1500x0.004 er1000x0.01 er1900x0.004; do python tools/run_general.py $c 100 20 256 2>/dev/null | grep -v path; done
bash tools/build_diag.sh 2>&1 | grep -i error
DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_big2.py mc1500 20 256
timeout 900 python -m pytest tests/test_gpu_big2.py -x -q -p no:cacheprovider 2>&1 | tail -2
