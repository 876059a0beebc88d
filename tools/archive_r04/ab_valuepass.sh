#!/bin/bash
# same-box A/B of the mixed mode's options: value pass on/off, input-gradient precision
for i in 1 2; do
for cfg in "1 f16" "0 f16"; do
  set -- $cfg
  v=$(VD_VALUE_PASS=$1 python bench.py --steps 20 --warmup 3 --prec-bwd $2 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f steps/s %.2f ms (median %.2f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))")
  echo "value_pass=$1 bwd=$2: $v"
done; done
