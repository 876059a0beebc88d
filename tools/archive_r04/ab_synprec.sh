cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_ab_synprec.txt; : > $OUT
for rep in 1 2; do for cfg in "f16x3 f16x3" "f16x3 f16" "f16 f16"; do set -- $cfg; python bench.py --prec-syn $1 --prec-bwd $2 --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 40 --warmup 5 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM syn fwd $1 bwd $2:', round(d['value'],3), round(d['ms_per_step'],3), [(p['program'],p['operands'],round(p['ms_total']/p['launches'],2)) for p in d['roofline']['programs'][:5]])" >> $OUT; done; done
cat $OUT
