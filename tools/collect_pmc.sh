#!/bin/bash
# Runs on the GPU box: SQ / LDS counters of k_fused (tools/run_fused.py er 5), one rocprofv3 pass per counter group
# (--pmc with --kernel-trace only, as the pool requires).  Output: gpurun_out/pmc/<group>/...counter_collection.csv
export TMPDIR=/tmp
GRAPHS=${1:-500}   # collect_pmc.sh 1 = one graph (the cluster variant of the kernel) into gpurun_out/pmc_1
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc
[ "$GRAPHS" != 500 ] && O=${O}_$GRAPHS
rm -rf "$O"; mkdir -p "$O"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_UNALIGNED_STALL" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/g$i" -- python3 tools/run_fused.py er 5 20 $GRAPHS > "$O/g$i.log" 2>&1
done
find "$O" -name "*counter_collection.csv" | head
