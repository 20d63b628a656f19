#!/bin/bash
# round 6, call 17: the suite three times over, logs kept whole - call 15's run ended in a core dump that call 16 did not reproduce
for i in 1 2 3; do
  timeout 2400 python -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider > gpurun_out/r06_suite_run$i.log 2>&1
  echo "run $i rc=$?: $(tail -1 gpurun_out/r06_suite_run$i.log | cut -c1-120)"
  grep -n "Fatal Python\|Memory access fault\|HSA_STATUS\|Aborted\|Segmentation" gpurun_out/r06_suite_run$i.log | head -5
done
dmesg 2>/dev/null | tail -5
