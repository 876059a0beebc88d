cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for p in 0 1; do
  export VD_PREP_STREAM=$p
  timeout 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_prep$p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs --no-alone > /dev/null 2>&1
done
ls $R/gpurun_out/trace_prep1/*/
