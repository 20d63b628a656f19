#!/bin/bash
# round 4, call 23: the multi-channel scheduling simulation on joint 3 x 300 conflict graphs, all five schedulers
mkdir -p gpurun_out
timeout 1500 python tools/run_wireless_mc.py 32 300 3 50 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_wireless_mc900.txt
cat gpurun_out/r04_wireless_mc900.txt
