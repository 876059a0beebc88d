#!/bin/bash
# Same-box A/B of environment switches on the default bench step: tools/ab_env.sh "VAR=a" "VAR=b" ... (each spec may hold several
# VAR=value pairs separated by semicolons; "-" = defaults); two alternating rounds, short runs without the extra legs.
for rep in 1 2; do
  for spec in "$@"; do
    ( if [ "$spec" != "-" ]; then IFS=';'; for kv in $spec; do export "$kv"; done; unset IFS; fi
      python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs --no-alone 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print('$spec', 'ms_per_step %.3f median %.3f steps/s %.2f' % (b['ms_per_step'], b['ms_per_step_median'], b['value']))" )
  done
done
