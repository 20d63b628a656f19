#!/bin/bash
# round 3, call 1: f64 MFMA semantics + rates; phase clocks and counters of the shallow configurations (C2, C4 l=1)
./tools/micro/mfma_f64 2>&1 | tee gpurun_out/r03_mfma_f64.txt
bash tools/build_diag.sh 2>&1 | tail -2
for cfg in "er100 1 500" "ba 1 500" "er200 1 500"; do
  echo "=== stamps $cfg"
  DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py $cfg 2>&1 | head -24
done 2>&1 | tee gpurun_out/r03_shallow_stamps.txt
python tools/run_fused.py er100 200 1 500
python tools/run_fused.py ba 200 1 500
bash tools/collect_pmc_cfg.sh c2 er100 1 500 > /dev/null
bash tools/collect_pmc_cfg.sh c4l1 ba 1 500 > /dev/null
ls gpurun_out/pmc_c2 gpurun_out/pmc_c4l1 | head -30
