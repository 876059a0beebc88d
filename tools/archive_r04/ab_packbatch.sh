cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_packbatch.txt; : > $OUT
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_wgrad.py tests/test_gpu_config_geometry.py tests/test_gpu_deterministic.py tests/test_gpu_embed.py -m gpu -q -x 2>&1 | grep "passed\|failed" | tail -2 >> $OUT
for rep in 1 2 3; do for v in 0 1; do VD_PACK_BATCH=$v python bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MTT pack_batch=$v', round(d['value'],3), round(d['ms_per_step'],3), d.get('grand_loss_last'))" >> $OUT; done; done
for v in 0 1; do VD_PACK_BATCH=$v python tools/bench_train.py 50 2>/dev/null | head -1 | sed "s/^/train pack_batch=$v: /" >> $OUT; done
for rep in 1 2; do for v in 0 1; do VD_PACK_BATCH=$v python bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DC pack_batch=$v', round(d['value'],3), round(d['ms_per_step'],2))" >> $OUT; done; done
cat $OUT
