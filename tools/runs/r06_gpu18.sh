#!/bin/bash
# round 6, call 18: the suite in the form of call 15 (-q, pytest's own faulthandler), four times, logs kept whole
for i in 1 2 3 4; do
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_suite_q$i.log 2>&1
  echo "run $i rc=$?: $(grep -E "passed|failed" gpurun_out/r06_suite_q$i.log | tail -1 | cut -c1-100)"
  grep -n "Fatal Python\|Memory access fault\|HSA_STATUS\|Aborted\|Segmentation\|core" gpurun_out/r06_suite_q$i.log | head -5
done
