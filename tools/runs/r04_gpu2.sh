#!/bin/bash
# round 4, call 2: whole GPU suite after the any-size path went into dgcn_solve_batch / dgcn_solve_residual_batch
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04_gpu2.log 2>&1
tail -5 gpurun_out/r04_gpu2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
