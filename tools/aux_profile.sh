#!/bin/bash
# The HBM-bound helpers only: rocprofv3 --kernel-trace --stats over tools/bench_aux.py -> gpurun_out/prof/aux_kernel_stats.csv + aux_kernels.json
# (the part of tools/refresh_profiles.sh that tools/aux_kernel_gbps.py reads; for a change that touches aux_kernels.hip alone)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/aux
timeout -k 5 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aux -- python3 $ROOT/tools/bench_aux.py > $OUT/aux.log 2>&1
cp $(ls $OUT/aux/*/*kernel_stats.csv | head -1) $OUT/aux_kernel_stats.csv
rm -rf $OUT/aux
cd $ROOT
python3 tools/aux_kernel_gbps.py $OUT/aux_kernel_stats.csv $OUT/aux_kernels.json
