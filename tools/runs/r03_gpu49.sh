#!/bin/bash
# round 3, call 49: the C5 search with batches that fill the chip (one graph per CU and two rounds)
for g in 64 128 256 512; do python tools/run_iterative.py --graphs $g --host 0 2>&1 | tail -1 | cut -c1-220; done
