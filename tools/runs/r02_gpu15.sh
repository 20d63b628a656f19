#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02n
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for i in 1 2; do
  for v in base pf own; do
    L=$R/distgcn_amd/libdgcn_$v.so; [ $v = base ] && L=$R/distgcn_amd/libdgcn.so
    DGCN_LIB=$L timeout 200 python tools/ab_fused.py "" 2>&1 | grep median | sed "s/^/$v /" >> "$O/ab.log"
  done
done
cat "$O/ab.log"
