#!/bin/bash
# Error / cost matrix of the shipped mode's two precision knobs: real side's last level x1 | x3, input gradient f16 | f16x3.
# For every combination: the late-regime parity test at 64x64x8 (summary line) and a short headline bench.  usage: tools/parity_matrix.sh [out dir]
out=${1:-gpurun_out/parity_matrix}
mkdir -p $out
for rl in x1 x3; do for bwd in f16 f16x3; do
  tag=${rl}_${bwd}
  VD_REAL_LAST=$rl VD_PARITY_BWD=$bwd VD_PARITY_STEPS=12 VD_PARITY_GRAD_BAR=1 VD_PARITY_LOG=$out/parity_$tag.json \
    python -m pytest tests/test_gpu_parity_late.py -q -s -k "oracle_64" 2>&1 | grep -E "shipped|x3  |passed|failed" > $out/parity_$tag.txt
  python bench.py --steps 12 --warmup 4 --real-last $rl --prec-bwd $bwd --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 > $out/bench_$tag.json 2> $out/bench_$tag.err
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f.split("bench_")[-1], "%.2f steps/s %.2f ms" % (d["value"], d["ms_per_step"]), "fwd1 %.2f ms" % r["mean_launch_ms"], {k: v for k, v in r.items() if k.startswith("fwd")})
PY
cat $out/parity_*.txt
