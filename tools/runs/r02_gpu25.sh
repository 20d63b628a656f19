#!/bin/bash
for k in 3 4 5 6 7 8; do echo "== forced K=$k"; timeout 300 python tools/cluster_check.py $k 2>&1 | tail -11; done
