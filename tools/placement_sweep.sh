#!/bin/bash
# Code-placement / scheduler sweep of fused.hip (the C3 launch moves by 4 % with its code placement): one libdgcn_<tag>.so per
# flag set under build/sweep/, every other object taken from build/obj as built by __graft_entry__.build().  tools/ab_libs.py times them.
#   tools/placement_sweep.sh tag "flags" [tag "flags" ...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/distgcn_amd/csrc"
mkdir -p "$ROOT/build/sweep"
OTHERS=""; for s in runtime pack supports supports2 spmm transform layer forward lgs shallow expand host_solver general big tail wide big2; do OTHERS="$OTHERS $ROOT/build/obj/$s.o"; done
while [ $# -ge 2 ]; do
  tag="$1"; flags="$2"; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c fused.hip -o "$ROOT/build/sweep/fused_$tag.o" &&
    hipcc --offload-arch=gfx950 -fPIC -shared -Wl,-z,defs -o "$ROOT/build/sweep/libdgcn_$tag.so" "$ROOT/build/sweep/fused_$tag.o" $OTHERS &&
    echo "built $tag: $flags" ) &
done
wait
