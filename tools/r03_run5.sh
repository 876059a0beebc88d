#!/bin/bash
python -m pytest tests/test_gpu_embed.py tests/test_gpu_cdriver.py tests/test_gpu_train.py tests/test_gpu_autograd.py -q 2>&1 | grep -E "FAILED|passed|failed|^E  " | cut -c1-300
for v in 0 1 0 1; do
  echo "VD_TAP_MASKS=$v"; VD_TAP_MASKS=$v python tools/perf_layers.py 512 f16,f16x3 2>&1 | grep -E "pix2slots"
done
for v in 0 1 0 1; do
  VD_TAP_MASKS=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs > gpurun_out/r03_mask_$v.json 2> gpurun_out/r03_mask_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_mask_$v.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("bench VD_TAP_MASKS=$v", "%.2f steps/s %.2f ms" % (d["value"], d["ms_per_step"]), "fwd1 %.2f ms" % r["mean_launch_ms"], {k: round(v, 1) for k, v in r.items() if k.startswith("fwd")})
PY
done
