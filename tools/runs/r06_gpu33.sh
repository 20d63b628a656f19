#!/bin/bash
# round 6, call 33: final state - whole GPU suite, the bench line un-profiled and under rocprofv3 (kernel stats)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_r06
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
cd /tmp
python3 $R/bench.py 2>/dev/null | tail -1 > $O/bench_default.unprofiled.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_default -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 6 --no-spmm-probe 2>/dev/null | tail -1 > $O/bench_c4_l20.unprofiled.json
ls -R $O | head -30
