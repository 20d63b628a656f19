#!/bin/bash
for c in mc1500 er1000x0.01 er1500x0.004; do python tools/run_general.py $c 100 20 256 2>/dev/null | grep -v path; done
