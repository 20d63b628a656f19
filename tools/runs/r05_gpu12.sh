#!/bin/bash
# round 5, call 12: where a search's GPU time goes on the any-size path (64 joint 3 x 300 graphs), plain k_big as a control
python tools/run_iterative.py --graphs 64 --family mc --n 900 --p 0.03 --layers 20 --host 0 2>/dev/null | grep -v path
for c in er500 mc900 er500 mc900; do python tools/run_general.py $c 300 20 256 2>/dev/null | grep big_solve; done
