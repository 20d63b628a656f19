#!/bin/bash
# several builds of the library on every BASELINE config, un-profiled, same box: bash tools/r02_gpu19.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
run() { python bench.py "$@" --cpu-seconds 0 --no-spmm-probe --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'], end='  ')"; }
for lib in "$@"; do
  export DGCN_LIB=$lib
  printf "%-34s" $lib
  echo -n "C2 "; run --nodes 100 --layers 1 --steps 1000 --warmup 20
  echo -n "C3 "; run --steps 1000 --warmup 20
  echo -n "C4l1 "; run --family ba --layers 1 --steps 500 --warmup 20
  echo -n "C4l20 "; run --family ba --layers 20 --steps 300 --warmup 20
  python tools/time_small.py 1 2>/dev/null | head -1 | awk '{printf "B1 %s us", $3}'
  echo
done
