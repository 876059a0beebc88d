#!/bin/bash
# Matrix-pipe / LDS / vector-memory utilisation counters of the three forward programs (separate --pmc passes,
# kernel-trace only).  Results in gpurun_out/pmc_sq/.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_l1.py 512 f16 > $OUT/p$i.log 2>&1
  cp $(ls $OUT/p$i/*/*counter_collection.csv | head -1) $OUT/pass$i.csv 2>/dev/null
  rm -rf $OUT/p$i
done
ls -la $OUT
