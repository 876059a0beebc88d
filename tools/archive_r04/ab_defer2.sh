cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_defer2.txt; : > $OUT
for rep in 1 2 3 4; do for cfg in "0 x3" "1 c8"; do set -- $cfg; VD_DEFER_BWD=$1 python bench.py --real-last $2 --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM defer=$1 real_last=$2', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
