#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and time of the fp8-corrected last level alone, for both box orders:
# VD_C8_BOXMAJOR=0 (clip-major: a workgroup's neighbours use all four window operand sets) and 1 (window-major: an XCD's resident
# workgroups share one set) -> gpurun_out/pmc_c8_traffic.txt
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_c8t; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for bm in 0 1; do
  export VD_C8_BOXMAJOR=$bm
  python3 $ROOT/tools/run_l2_c8.py 3200 6 2>/dev/null | tail -1 > $OUT/plain_$bm.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_${bm}_$c -- python3 $ROOT/tools/run_l2_c8.py 3200 4 > $OUT/p_${bm}_$c.log 2>&1
    cp $(ls $OUT/p_${bm}_$c/*/*counter_collection.csv | head -1) $OUT/${c}_$bm.csv 2>/dev/null
    rm -rf $OUT/p_${bm}_$c
  done
done
python3 - <<PY > $ROOT/gpurun_out/pmc_c8_traffic.txt
import csv
algo = 3200 * (2 * 128 * 8 * 7 * 7 * 2 + 2048 * 4)
for bm in (0, 1):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = [float(r["Counter_Value"]) for r in csv.DictReader(open("$OUT/%s_%d.csv" % (c, bm))) if "conv_mfma_kernel<4" in r["Kernel_Name"] and r["Counter_Name"] == c]
        v[c] = sum(rows[1:]) / max(1, len(rows[1:]))
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    print("VD_C8_BOXMAJOR=%d: HBM bytes per 3200-clip launch %.3f GB = %.2fx of %.3f GB algorithmic;  %s" % (bm, hbm / 1e9, hbm / algo, algo / 1e9, open("$OUT/plain_%d.log" % bm).read().strip()))
PY
cat $ROOT/gpurun_out/pmc_c8_traffic.txt
