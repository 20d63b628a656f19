#!/bin/bash
# round 4, call 40: k_big per hidden layer and fixed part (ER500 / MC900 at 3, 4, 12, 20 layers, random weights)
for c in er500 mc900; do for l in 3 4 12 20; do
  python tools/run_general.py $c 200 $l 256 2>/dev/null | grep big_solve
done; done
