#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02i
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for i in 1 2 3; do
  timeout 300 python tools/tune_spmm_hbm.py 2>&1 | grep "{}" >> "$O/new.log"
  DGCN_LIB=$R/distgcn_amd/libdgcn_oldstage.so timeout 300 python tools/tune_spmm_hbm.py 2>&1 | grep "{}" >> "$O/old.log"
done
echo NEW; cat "$O/new.log"; echo OLD; cat "$O/old.log"
