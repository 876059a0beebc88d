cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ab_c8pos.txt; : > $OUT
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 >> $OUT
# the real side alone: kernel durations of the last level, position tiles against the row-major program
cd /tmp; export TMPDIR=/tmp
for p in 0 1; do
  export VD_C8_POS=$p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c8pos$p -- python3 $GRAFT_REPO_ROOT/tools/run_real_side.py 4 > /dev/null 2>&1
  echo "== real side alone, VD_C8_POS=$p" >> $OUT
  python3 - >> $OUT <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c8pos$p/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("  ", r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for p in 0 1; do VD_C8_POS=$p python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM c8_pos=$p', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
