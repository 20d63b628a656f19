#!/bin/bash
# round 3, call 47: step time against remaining vertices (product build, one event pair per step)
python tools/step_curve.py rollout 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_step_curve.txt
python tools/step_curve.py cit 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/r03_step_curve.txt
