#!/bin/bash
# round 6, call 43: k_graph_rank with sixteen lanes per graph - the order tests, then its duration on the C4 share / C4 / mixed ER under rocprofv3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rank16
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -x -q -k "dispatch or order or random_shapes_match" 2>&1 | tail -2 | tee $O/tests.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_share -- python3 $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 0 > $O/c4_share.json 2> $O/c4_share.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_full -- python3 $R/bench.py --config C4 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 0 > $O/c4_full.json 2> $O/c4_full.err
for d in c4_share c4_full; do head -3 $O/$d/runc/*_kernel_stats.csv | cut -c1-150; done
python3 $R/bench.py --config C4 --cpu-seconds 0 --no-spmm-probe --no-e2e 2>/dev/null | tail -1 > $O/c4_full.unprofiled.json
python3 $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 0 --no-spmm-probe --no-e2e 2>/dev/null | tail -1 > $O/c4_share.unprofiled.json
cd $R; DGCN_AB_KIND=ermix python tools/ab_fused.py "" 2>&1 | tail -1
