#!/bin/bash
# rocprofv3 kernel stats of one command: tools/prof_one.sh NAME python3 <script> [args]  -> gpurun_out/prof2/NAME_kernel_stats.csv
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof2; mkdir -p $OUT; export TMPDIR=/tmp
name=$1; shift
cmd=("$@"); for i in "${!cmd[@]}"; do [ -e "$ROOT/${cmd[$i]}" ] && cmd[$i]="$ROOT/${cmd[$i]}"; done
cd /tmp
timeout -k 5 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "${cmd[@]}" > $OUT/$name.log 2>&1
cp $(ls $OUT/$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv; rm -rf $OUT/$name
