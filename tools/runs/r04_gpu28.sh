#!/bin/bash
# round 4, call 28: mixed batch (C4 share) as size classes on concurrent streams against one launch
python tools/class_split_probe.py > gpurun_out/r04_gpu28.log 2>&1
cat gpurun_out/r04_gpu28.log
