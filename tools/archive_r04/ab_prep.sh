cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_prep.txt; : > $OUT
timeout 600 python -m pytest tests/test_gpu_c8.py tests/test_gpu_parity_late.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
for rep in 1 2 3 4; do for p in 0 1; do VD_PREP_STREAM=$p python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM prep_stream=$p', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
