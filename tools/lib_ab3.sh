#!/bin/bash
# same-box A/B of libvd_hip_old.so vs libvd_hip.so on the x3-heavy paths: train step, MTT+Ours iteration, DC step, DM step
OLD=$PWD/video_distillation_amd/libvd_hip_old.so
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export VD_LIB_PATH=$OLD; else unset VD_LIB_PATH; fi
    echo -n "$v train: "; python tools/train_host_time.py 50 2>&1 | tail -1
    echo -n "$v mtt:   "; python tools/mtt_host_time.py 2>&1 | tail -1
    echo -n "$v dc:    "; python bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 --sustain-seconds 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f steps/s %.1f ms' % (d['value'], d['ms_per_step']))"
    echo -n "$v dm:    "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f steps/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
