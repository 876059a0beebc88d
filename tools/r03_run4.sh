#!/bin/bash
python -m pytest tests/test_gpu_embed.py -q -s -k "variants_are_bitwise" 2>&1 | grep -E "FAILED|passed|failed|^E  " | cut -c1-300
for rep in 1 2; do for v in 3 4; do
  echo "VD_L0_BREG=$v (rep $rep)"; VD_L0_BREG=$v python tools/perf_layers.py 512 f16 2>&1 | grep -E "pix2slots"
done; done
for v in 3 4 3 4; do
  VD_L0_BREG=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs > gpurun_out/r03_breg_$v.json 2> gpurun_out/r03_breg_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_breg_$v.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("bench VD_L0_BREG=$v", "%.2f steps/s %.2f ms" % (d["value"], d["ms_per_step"]), "fwd1 %.2f ms" % r["mean_launch_ms"], {k: round(v, 1) for k, v in r.items() if k.startswith("fwd")})
PY
done
