#!/bin/bash
# round 3, call 2: output layout + chain semantics of v_mfma_f64_16x16x4_f64
./tools/micro/mfma_f64 2>&1 | head -5 | tee gpurun_out/r03_mfma_f64_layout.txt
