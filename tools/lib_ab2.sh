#!/bin/bash
# same-box A/B: DM bench and DC bench, old library (libvd_hip_old.so) vs current, alternating
run() { python bench.py "$@" --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f steps/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo -n "dm old: "; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so run --steps 20 --warmup 3
  echo -n "dm new: "; run --steps 20 --warmup 3
done
echo -n "dc old: "; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so run --method dc --classes 51 --ipc 5 --steps 3 --warmup 1
echo -n "dc new: "; run --method dc --classes 51 --ipc 5 --steps 3 --warmup 1
echo -n "mtt old: "; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so run --method mtt --classes 400 --frames 8 --size 64 --steps 4 --warmup 2
echo -n "mtt new: "; run --method mtt --classes 400 --frames 8 --size 64 --steps 4 --warmup 2
echo -n "dc old: "; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so run --method dc --classes 51 --ipc 5 --steps 3 --warmup 1
echo -n "dc new: "; run --method dc --classes 51 --ipc 5 --steps 3 --warmup 1
