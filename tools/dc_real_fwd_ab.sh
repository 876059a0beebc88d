#!/bin/bash
# DC (config 4) with the real batch's forward in hi+lo pairs (default) vs single-pass f16 (optionally with dithered weight sets):
# the class-term test's errors and the bench line.   usage: tools/dc_real_fwd_ab.sh "f16x3:0 f16:0 f16:8" [nobench]
OUT=gpurun_out/dc_ab; mkdir -p $OUT
for spec in ${1:-f16x3:0 f16:0 f16:8}; do
  mode=${spec%%:*}; dith=${spec##*:}
  VD_TRAIN_DITHER=$dith VD_PREC_TRAIN=$mode timeout 900 python3 -m pytest tests/test_gpu_config_geometry.py -m gpu -q -s -k config4 2>&1 | grep -E "config-4 geometry: |passed|failed|Error" > $OUT/test_${mode}_$dith.txt
  echo "== $mode dither $dith"; cat $OUT/test_${mode}_$dith.txt
  [ -n "$2" ] && continue
  VD_TRAIN_DITHER=$dith VD_PREC_TRAIN=$mode timeout 900 python3 bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${mode}_$dith.json
  python3 -c "
import json; d=json.load(open('$OUT/bench_${mode}_$dith.json')); print(d['value'], d['ms_per_step']); print([(p['program'],p['operands'],round(p['ms_total'],1)) for p in d['roofline']['programs']])"
done
