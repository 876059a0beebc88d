# same-box A/B of the tiny-launch rule of plan.latency_variant (VD_TINY_GRID=0: off) on the loops that issue small launches
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_tiny.txt; : > $OUT
for w in 0 128 0 128; do VD_TINY_GRID=$w python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 30 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM tiny=$w', d['value'], d['ms_per_step'], d['ms_per_step_median'])" >> $OUT; done
for w in 0 128; do VD_TINY_GRID=$w python bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DC tiny=$w', d['value'], d['ms_per_step'])" >> $OUT; done
for w in 0 128; do VD_TINY_GRID=$w python bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MTT tiny=$w', d['value'], d['ms_per_step'])" >> $OUT; done
for w in 0 128; do VD_TINY_GRID=$w python tools/bench_train.py 50 2>/dev/null | head -1 | sed "s/^/train tiny=$w /" >> $OUT; done
cat $OUT
