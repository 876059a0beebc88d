#!/bin/bash
# SQ / TA counters of the fp8-corrected last level alone (separate --pmc passes, kernel-trace only) -> gpurun_out/pmc_c8/
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_c8; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python3 $ROOT/tools/run_l2_c8.py 3200 5 > $OUT/plain.log 2>&1
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_l2_c8.py 3200 3 > $OUT/p$i.log 2>&1
  cp $(ls $OUT/p$i/*/*counter_collection.csv | head -1) $OUT/pass$i.csv 2>/dev/null
  rm -rf $OUT/p$i
done
cat $OUT/plain.log | tail -2
python3 - <<PY
import csv, glob
from collections import defaultdict
for f in sorted(glob.glob("$OUT/pass*.csv")):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_mfma_kernel<4" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    print(f.split("/")[-1], {k: round(sum(v) / len(v), 1) for k, v in acc.items()})
PY
