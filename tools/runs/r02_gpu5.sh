#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "wide or shape_coverage or every_shipped or full_size or margin or edge_case or random_graphs or heads or exploring or c5_rollout_full or serving" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/summary.txt"
for i in 1 2; do
  timeout 200 python tools/ab_fused.py "" >> "$O/ab_valu.log" 2>&1
  DGCN_LIB=$R/distgcn_amd/libdgcn_novalu.so timeout 200 python tools/ab_fused.py "" >> "$O/ab_novalu.log" 2>&1
done
cat "$O/summary.txt"; tail -5 "$O/pytest.log"; echo VALU; grep median "$O/ab_valu.log"; echo NOVALU; grep median "$O/ab_novalu.log"
