#!/bin/bash
# usage: tools/build_variant.sh <conv_mfma source> <out .so> [extra hipcc flags]   -- a library with another build of the conv kernel, the
# other objects as built in csrc/build (same-box A/B through VD_LIB_PATH)
SRC=$1; OUT=$2; shift 2
D=video_distillation_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC "$@" -I$D -c $SRC -o /tmp/variant_conv_$$.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/variant_conv_$$.o $(ls $D/build/*.o | grep -v conv_mfma) -ldl
