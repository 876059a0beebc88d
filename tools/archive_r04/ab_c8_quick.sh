cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_c8.py -q -s 2>&1 | grep -a "per-clip\|passed\|failed"
for m in x3 c8 x3 c8; do python bench.py --real-last $m --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('DM real_last=$m', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'alone', {k:round(v,3) for k,v in r['alone'].items() if k.endswith('_ms')})"; done
