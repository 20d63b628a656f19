#!/bin/bash
# evidence for the single-graph latency numbers in DESIGN.md
mkdir -p gpurun_out
{
echo "# one solve_mwis call, N = 200 ER graph, 20 layers (tools/run_single.py 1000)"
python tools/run_single.py 1000 | tail -4
echo
echo "# where the call's time goes (tools/lat_probe.py)"
python tools/lat_probe.py 1000 | tail -4
echo
echo "# kernel alone: one workgroup per graph vs cluster variant, automatic K (tools/cluster_check.py)"
python tools/cluster_check.py | grep -v "^$"
} > gpurun_out/r02_single_graph_latency.txt 2>&1
tail -50 gpurun_out/r02_single_graph_latency.txt
