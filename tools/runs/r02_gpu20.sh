#!/bin/bash
# kernel-only time (dgcn_timing) of the C2 / C4-l1 configs, old against new library
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  export DGCN_LIB=$lib
  printf "%-34s" $lib
  for cfg in "--nodes 100 --layers 1" "--family ba --layers 1"; do
    python bench.py $cfg --steps 1000 --warmup 20 --cpu-seconds 0 --no-spmm-probe --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step %.2f us kernel %.2f us' % (d['ms_per_step']*1e3, d['kernels']['fused_solve']['avg_us']), end='   ')"
  done
  echo
done
