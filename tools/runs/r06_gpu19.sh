#!/bin/bash
# round 6, call 19: this round's new tests in a loop (fresh process each time) - is any of them the source of call 15's core dump?
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "three_support or narrow or cluster_fault or big2_iterative_solvers_restatement or two_layer_iterative_solvers_restatement or hub_rows or chebyshev or supports2" > gpurun_out/r06_loop$i.log 2>&1
  echo "loop $i rc=$?: $(grep -E "passed|failed" gpurun_out/r06_loop$i.log | tail -1 | cut -c1-100)"
  grep -n "Fatal Python\|Memory access fault\|Aborted\|Segmentation\|core" gpurun_out/r06_loop$i.log | head -3
done
