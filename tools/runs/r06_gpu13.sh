#!/bin/bash
# round 6, call 13: the product build in the A/B harness of the layer-loop experiment (tools/ab_fused.py), for reference
python tools/ab_fused.py "" "fused_block=1024" 2>&1 | tail -2 | tee gpurun_out/r06_ab_product.txt
