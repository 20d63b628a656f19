#!/bin/bash
# The -DDGCN_DIAG profiling build of the library (phase clocks / ablation switches in k_fused; tools/stamp_fused.py).
# Not part of the product: built on demand, selected with DGCN_LIB=distgcn_amd/libdgcn_diag.so.
cd "$(dirname "$0")/../distgcn_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDGCN_DIAG -Wl,-z,defs \
  -o ../libdgcn_diag.so runtime.hip pack.hip supports.hip supports2.hip spmm.hip transform.hip layer.hip forward.hip lgs.hip fused.hip \
  shallow.hip expand.hip host_solver.hip general.hip big.hip tail.hip wide.hip big2.hip
