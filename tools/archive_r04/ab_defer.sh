# same-box A/B: real side's last level x3 / c8 (fp8 corrections), with and without the deferred backward of the synthetic side
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_defer.txt; : > $OUT
for rep in 1 2 3; do for d in 0 1; do for m in x3 c8; do VD_DEFER_BWD=$d python bench.py --real-last $m --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('DM defer=$d real_last=$m', round(d['value'],3), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), {p['program']: round(p['ms_total']/p['launches'],2) for p in r['programs'] if p['program'] in ('fwd0','fwd1','fwd2_hilo','fwd2_c8') and p['operands'] in ('f16','f16x3','f16c8')})" >> $OUT; done; done; done
python tools/mfma_peak.py >> $OUT 2>/dev/null
cat $OUT
