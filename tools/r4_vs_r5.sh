#!/bin/bash
# same-box round 4 (tree at 1ebd716 under .r04tree) vs HEAD: three alternating pairs of short default bench runs
R=$(pwd)   # (build the tree first, in the build container: git worktree add -f .r04tree 1ebd716 && (cd .r04tree && python -c "from video_distillation_amd import hip; hip.build()"); it travels with the gpurun snapshot; remove it afterwards)
run() { ( cd $1; python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs 2>/dev/null | tail -1 | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); a=b['roofline'].get('alone') or {}; print('$2', 'steps/s %.2f  ms %.3f  median %.3f  alone l0 %.2f l1 %.2f l2 %.2f  in-step fwd1 %.2f' % (b['value'], b['ms_per_step'], b['ms_per_step_median'], a.get('fwd0_ms',0), a.get('fwd1_ms',0), a.get('fwd2_c8_ms',0), b['roofline']['mean_launch_ms']))" ) }
for rep in 1 2 3; do run $R/.r04tree "round 4 (1ebd716)"; run $R "round 5 (HEAD)   "; done
