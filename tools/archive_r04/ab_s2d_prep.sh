cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_s2d_prep.txt; : > $OUT
for rep in 1 2 3; do for v in 0 1; do VD_PREP_STREAM=$v python bench.py --method s2d --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s2d prep_stream=$v', round(d['value'],3), round(d['ms_per_step'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
