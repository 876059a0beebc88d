# same-box A/B of the real side's last level: x3 (three fp16 MFMAs per product) vs c8 (fp16 + fp8 corrections), and the parity tests under c8
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_c8.txt; : > $OUT
for m in x3 c8 x3 c8; do python bench.py --real-last $m --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('DM real_last=$m', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'L2 ms in-step', round(r.get('fwd2_ms_per_launch',0),3), 'alone', {k:round(v,3) for k,v in r['alone'].items() if k.endswith('_ms')}, 'loss', d['loss_last'])" >> $OUT; done
VD_REAL_LAST=c8 VD_PARITY_LOG=gpurun_out/r04_parity_c8.json python -m pytest tests/test_gpu_parity_late.py -q -s 2>&1 | grep -a "shipped  clean\|shipped  vs oracle\|passed\|failed" >> $OUT
cat $OUT
