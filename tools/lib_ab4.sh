#!/bin/bash
# same-box A/B of libvd_hip_old.so vs libvd_hip.so: stand-alone forward layers (512 clips) and the DM bench
OLD=$PWD/video_distillation_amd/libvd_hip_old.so
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export VD_LIB_PATH=$OLD; else unset VD_LIB_PATH; fi
    echo -n "$v layers: "; python tools/perf_layers.py 512 f16 2>&1 | tail -1
    echo -n "$v dm:     "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f steps/s  %.2f ms frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
  done
done
