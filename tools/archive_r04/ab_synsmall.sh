cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_synsmall.txt; : > $OUT
for rep in 1 2 3; do for v in 0 1; do VD_SYN_BWD0_SMALL=$v python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM syn_bwd0_small=$v', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
for rep in 1 2; do for v in 0 1; do VD_HEAD_GATHER=$v python bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MTT head_gather=$v', round(d['value'],3), round(d['ms_per_step'],3))" >> $OUT; done; done
for v in 0 1; do VD_HEAD_GATHER=$v python tools/bench_train.py 50 2>/dev/null | tail -2 | sed "s/^/train head_gather=$v: /" >> $OUT; done
cat $OUT
