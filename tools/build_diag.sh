#!/bin/bash
# The -DDGCN_DIAG profiling build of the library (phase clocks / ablation switches in k_fused; tools/stamp_fused.py).
# Not part of the product: built on demand (here or on the GPU box: hipcc cross-compiles), selected with
# DGCN_LIB=distgcn_amd/libdgcn_diag.so.  One object per source, side by side, like __graft_entry__.build().
set -e
cd "$(dirname "$0")/../distgcn_amd/csrc"
mkdir -p ../../build/obj_diag
SRCS="runtime pack supports supports2 spmm transform layer forward lgs fused shallow expand host_solver general big tail wide big2"
for s in $SRCS; do
  if [ ! -f ../../build/obj_diag/$s.o ] || [ -n "$(find $s.hip *.h ../../include/dgcn.h -newer ../../build/obj_diag/$s.o 2>/dev/null | head -1)" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDGCN_DIAG -c $s.hip -o ../../build/obj_diag/$s.o &
  fi
done
wait
OBJS=""; for s in $SRCS; do OBJS="$OBJS ../../build/obj_diag/$s.o"; done
hipcc --offload-arch=gfx950 -fPIC -shared -Wl,-z,defs -o ../libdgcn_diag.so $OBJS
