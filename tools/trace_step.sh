#!/bin/bash
# One kernel trace of the default bench step (no counters) + the per-stream timeline of a steady-state step -> gpurun_out/<tag>_timeline.txt
# usage (under gpurun): tools/trace_step.sh [tag] [extra bench.py arguments, e.g. --frames 8 --size 64]
TAG=${1:-r06}
shift
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$TAG
timeout 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs --no-alone "$@" > /dev/null 2>&1
F=$(ls $R/gpurun_out/trace_$TAG/*/*kernel_trace.csv | head -1)
cd $R
python3 tools/trace_timeline.py $F 2 > gpurun_out/${TAG}_timeline.txt
rm -rf gpurun_out/trace_$TAG
