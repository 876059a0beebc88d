cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ab_c8pos2.txt; : > $OUT
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12 >> $OUT
cd /tmp; export TMPDIR=/tmp
for bm in 0 1; do
  export VD_C8_BOXMAJOR=$bm
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c8bm$bm -- python3 $GRAFT_REPO_ROOT/tools/run_real_side.py 6 > /dev/null 2>&1
  echo "== real side alone, position tiles, VD_C8_BOXMAJOR=$bm" >> $OUT
  python3 - >> $OUT <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c8bm$bm/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print("  ", r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us  min", round(float(r["MinNs"])/1e3,1))
PY
done
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for bm in 0 1; do VD_C8_BOXMAJOR=$bm python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM boxmajor=$bm', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
