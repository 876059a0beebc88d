cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_cfg1_c8.txt; : > $OUT
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -14 >> $OUT
for rep in 1 2 3; do for v in x3 c8; do python bench.py --frames 8 --size 64 --real-last $v --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 60 --warmup 10 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config-1 shape real_last=$v', d['config']['precision']['real_clips_last_level'], d['config']['precision'].get('real_clips_last_level_program'), round(d['value'],2), round(d['ms_per_step'],3), d['loss_last'])" >> $OUT; done; done
VD_PARITY_STEPS=16 timeout 900 python -m pytest tests/test_gpu_parity_late.py -m gpu -q -x -s -k oracle_64 2>&1 | grep "shipped\|passed\|failed" >> $OUT
cat $OUT
