cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ab_lib3.txt; : > $OUT
L=$GRAFT_REPO_ROOT/video_distillation_amd
for rep in 1 2 3; do for v in A C hip; do
  echo "lib $v: $(VD_LIB_PATH=$L/libvd_$v.so timeout 300 python tools/run_l2_c8.py 3200 8 2>&1 | tail -1 | sed 's/.*launch: //')" >> $OUT
done; done
for rep in 1 2 3; do for v in A C hip; do VD_LIB_PATH=$L/libvd_$v.so python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM lib $v', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['loss_last'])" >> $OUT; done; done
cat $OUT
