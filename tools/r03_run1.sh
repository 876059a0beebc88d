#!/bin/bash
# round-3 GPU batch: re-run of the tests changed after the first full-suite run, then the syn-after-L0 A/B and the default bench
python -m pytest tests/test_gpu_parity_late.py tests/test_gpu_dither.py tests/test_gpu_collectives.py tests/test_gpu_e2e.py -q -s 2>&1 | grep -v "^$" > gpurun_out/r03_t3.log
grep -E "FAILED|passed|failed|DM 12 steps|evaluate_synset \(|shipped  vs|clean entries|entries with" gpurun_out/r03_t3.log | cut -c1-400
for v in 0 1; do
  VD_SYN_AFTER_L0=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs > gpurun_out/r03_synl0_$v.json 2> gpurun_out/r03_synl0_$v.err
done
python bench.py --steps 20 --warmup 5 --eval-epochs 30 > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
python - <<PY
import json
for f in ("r03_synl0_0", "r03_synl0_1", "r03_bench_a"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r = d["roofline"]
    print(f, "%.2f steps/s %.2f ms (median %.2f)" % (d["value"], d["ms_per_step"], d["ms_per_step_median"]), "fwd1 %.2f ms" % r["mean_launch_ms"],
          {k: round(v, 1) for k, v in r.items() if k.startswith("fwd")}, d.get("eval"), {k: (round(d[k]["value"], 2), round(d[k]["ms_per_step"], 2)) for k in ("parity_mode", "fast_mode") if k in d})
PY
