cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for m in x3 c8; do
  timeout 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$m -- python3 $R/bench.py --real-last $m --steps 8 --warmup 2 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs --no-alone > /dev/null 2>&1
done
ls $R/gpurun_out/trace_c8/*/
