#!/bin/bash
# A/B builds of the first-level kernel's tuning switches on one box: tools/l0_variants.sh "TAG:-DFLAG=.. -DFLAG=.." ... (build HERE, then run
# under gpurun: VARIANTS="tag tag" tools/l0_variants.sh --run [clips] [reps])
if [ "$1" == "--run" ]; then
  for rep in 1 2; do
    for t in default $VARIANTS; do
      if [ $t == default ]; then unset VD_LIB_PATH; else export VD_LIB_PATH=$PWD/video_distillation_amd/libvd_hip_$t.so; fi
      echo -n "$t: "; python tools/l0_ab.py ${2:-3200} ${3:-6} 5 2>&1 | grep 'BREG=5'
    done
  done
  exit 0
fi
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  tools/build_variant.sh video_distillation_amd/csrc/conv_mfma.hip video_distillation_amd/libvd_hip_$tag.so $flags &
done
wait
ls -la video_distillation_amd/libvd_hip_*.so
