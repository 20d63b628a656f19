#!/bin/bash
# round 6, call 16: which test brought the suite down in call 15 (verbose log kept whole)
timeout 2400 python -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider > gpurun_out/r06_suite_verbose.log 2>&1
echo "rc=$?"; grep -n "Fatal\|fault\|Aborted\|core" gpurun_out/r06_suite_verbose.log | head; tail -5 gpurun_out/r06_suite_verbose.log | cut -c1-300
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r06_suite_verbose.log | tail -3
