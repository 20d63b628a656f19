#!/bin/bash
# round 5, call 20: two-layer models in wide.hip: tests; timing against the layer-by-layer chain
timeout 1500 python -m pytest tests/test_gpu_wide.py -x -q --tb=short -p no:cacheprovider -k "two_layer" 2>&1 | tail -6
for w in 1 0; do for c in mc900 er1500x0.01; do echo -n "DGCN_WIDE2=$w $c l=2: "; DGCN_WIDE2=$w python tools/run_general.py $c 300 2 256 2>/dev/null | grep -v path | tr '\n' ';'; echo; done; done
