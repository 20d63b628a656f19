#!/bin/bash
# Like collect_pmc.sh, for any run_fused.py configuration: collect_pmc_cfg.sh <tag> <kind> <layers> <graphs>
# (one rocprofv3 pass per counter group; --pmc with --kernel-trace only).  Output: gpurun_out/pmc_<tag>/
export TMPDIR=/tmp
TAG=$1; KIND=$2; LAYERS=$3; GRAPHS=$4
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/g$i" -- python3 tools/run_fused.py $KIND 5 $LAYERS $GRAPHS > "$O/g$i.log" 2>&1
done
find "$O" -name "*counter_collection.csv" | head -20
