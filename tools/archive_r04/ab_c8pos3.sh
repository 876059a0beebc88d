cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ab_c8pos3.txt; : > $OUT
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12 >> $OUT
timeout 300 python tools/run_l2_c8.py 3200 6 2>&1 | tail -1 >> $OUT
timeout 300 python tools/ablate_c8.py 3200 2>&1 | tail -14 >> $OUT
for rep in 1 2; do python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done
cat $OUT
