#!/bin/bash
# round 6, call 42: rocprofv3 kernel stats for the C4 share (500 BA graphs, l = 20) and C4 on one GPU (4 000 graphs), C5
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c4_stats
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_share -- python3 $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 0 > $O/c4_share.json 2> $O/c4_share.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_full -- python3 $R/bench.py --config C4 --cpu-seconds 0 --no-spmm-probe --no-e2e --parity-seconds 0 > $O/c4_full.json 2> $O/c4_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 $R/bench.py --config C5 --cpu-seconds 0 > $O/c5.json 2> $O/c5.err
python3 $R/bench.py --config C5 --cpu-seconds 0 2>/dev/null | tail -1 > $O/c5.unprofiled.json
ls -R $O | head -40
