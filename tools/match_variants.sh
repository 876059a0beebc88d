#!/bin/bash
# same-box timing of vd_match_rows_* builds (video_distillation_amd/libvd_match_<X>.so, built by hand with -D switches): kernel averages of
# tools/bench_aux.py under rocprofv3
for v in "" "$@"; do
  if [ -n "$v" ]; then export VD_LIB_PATH=$GRAFT_REPO_ROOT/video_distillation_amd/libvd_match_$v.so; else unset VD_LIB_PATH; fi
  echo "== ${v:-shipped}"; tools/aux_profile.sh | grep match
done
