#!/bin/bash
# same-box A/B of two builds of the library: video_distillation_amd/libvd_hip_old.so (built by hand from an older
# conv_mfma.hip) vs the current libvd_hip.so -- first-layer kernel stand-alone and the whole DM bench
for i in 1 2; do
  echo "old:"; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so timeout 200 python tools/l0_knob_ab.py 512 0 2>&1 | grep -v amdgpu
  echo "new:"; timeout 200 python tools/l0_knob_ab.py 512 0 2>&1 | grep -v amdgpu
done
for i in 1 2; do
  echo "old:"; VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_old.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f steps/s  %.2f ms  fwd0 %.0f TF' % (d['value'], d['ms_per_step'], d['roofline'].get('fwd0_tflops', 0)))"
  echo "new:"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f steps/s  %.2f ms  fwd0 %.0f TF' % (d['value'], d['ms_per_step'], d['roofline'].get('fwd0_tflops', 0)))"
done
