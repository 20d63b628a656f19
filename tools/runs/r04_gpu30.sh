#!/bin/bash
# round 4, call 30: per-step cost of the tail kernel by model depth
python tools/tail_probe.py > gpurun_out/r04_gpu30.log 2>&1
cat gpurun_out/r04_gpu30.log
