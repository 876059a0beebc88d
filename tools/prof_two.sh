ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
prof() { local name=$1; shift
  timeout -k 5 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  cp $(ls $OUT/$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv; rm -rf $OUT/$name; }
VD_SKIP_TORCH=1 prof train python3 $ROOT/tools/train_host_time.py 50
VD_GM_LANES=1 prof dc1 python3 $ROOT/bench.py --method dc --classes 8 --ipc 5 --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0
ls $OUT
cd /tmp
prof mtt python3 $ROOT/bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0
ls $OUT
