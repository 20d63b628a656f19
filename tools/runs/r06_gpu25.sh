#!/bin/bash
# round 6, call 25: code-placement / scheduler-flag sweep of fused.hip on C3 (tools/placement_sweep.sh builds, tools/ab_libs.py times)
python tools/ab_libs.py distgcn_amd/libdgcn.so build/sweep/libdgcn_base.so build/sweep/libdgcn_noal.so build/sweep/libdgcn_nft5.so \
  build/sweep/libdgcn_nft6.so build/sweep/libdgcn_ilp.so build/sweep/libdgcn_nopost.so build/sweep/libdgcn_O2.so build/sweep/libdgcn_blk4.so \
  build/sweep/libdgcn_fn12.so build/sweep/libdgcn_bias50.so 2>&1 | tee gpurun_out/r06_placement_sweep.txt
