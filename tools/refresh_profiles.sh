#!/bin/bash
# Regenerate the evidence under profiles/ on a GPU box (run from the repo root through gpurun);
# results land in gpurun_out/prof/ and are copied to profiles/r06_* afterwards (tools/copy_profiles.sh).
# Every rocprofv3 run is bounded by `timeout` and writes csv (the rocpd default has hung a box for its whole limit).
# PMC passes run alone (--kernel-trace only next to --pmc), one counter group per run.
set -u
# PART=1: bench lines, same-box A/Bs, stand-alone timings; PART=2: rocprofv3 kernel stats, PMC passes, timeline (default: both).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
PART=${PART:-12}
export TMPDIR=/tmp
if [[ $PART == *1* ]]; then
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_1gpu.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --sustain-seconds 0 2>/dev/null | tail -1 > $OUT/bench_1gpu_again.json
python3 bench.py --method s2d --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_s2d.json
python3 bench.py --frames 8 --size 64 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_config1_shape.json
python3 bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_dc.json
python3 bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_mtt.json
VD_PREC_MATCH=bf16x3 python3 bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_mtt_bf16x3.json
VD_PREC_MATCH=bf16x3 python3 bench.py --method dc --classes 51 --ipc 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_dc_bf16x3.json
VD_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --eval-epochs 1 --eval-seeds 1 --exchange-leg --no-extra-legs 2>/dev/null | tail -1 > $OUT/bench_8ranks_one_device.json
python3 tools/mfma_peak.py > $OUT/mfma_peak.txt 2>/dev/null
python3 tools/l0_ab.py 3200 8 > $OUT/l0_kernel_ab.txt 2>/dev/null
python3 tools/stamps_l0.py 3200 5 2>/dev/null | tail -10 > $OUT/l0_phase_stamps.txt
python3 tools/hal_check.py 50 16 112 112 2>/dev/null | tail -2 > $OUT/hal_bwd.txt
VD_HAL_FUSED=0 python3 tools/hal_check.py 50 16 112 112 2>/dev/null | tail -2 >> $OUT/hal_bwd.txt
for n in 1 2 4 8; do python3 tools/rank_proxy.py $n 2>/dev/null; done > $OUT/rank_proxy.txt       # (one process per N: x of N=1 from the first line)
python3 tools/bench_train.py 50 > $OUT/train_step.txt 2>/dev/null
VD_DETERMINISTIC=1 python3 tools/bench_train.py 50 > $OUT/train_step_deterministic.txt 2>/dev/null
tools/micro/mfma_round_probe > $OUT/mfma_rounding.txt 2>&1
fi
[[ $PART == *2* ]] || { ls -la $OUT; exit 0; }
mkdir -p $OUT
cd /tmp
prof() {  # name, then the program and its arguments
  local name=$1; shift
  timeout -k 5 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  cp $(ls $OUT/$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv
}
prof bench python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --eval-epochs 0 --sustain-seconds 0 --no-extra-legs --no-alone
grep '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_1gpu_under_rocprof.json
cp $(ls $OUT/bench/*/*kernel_trace.csv | head -1) $OUT/bench_kernel_trace.csv
prof s2d python3 $ROOT/bench.py --method s2d --steps 8 --warmup 2 --no-cpu-baseline --sustain-seconds 0
prof dc python3 $ROOT/bench.py --method dc --classes 8 --ipc 5 --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0
prof mtt python3 $ROOT/bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0
prof aux python3 $ROOT/tools/bench_aux.py
prof syn_side python3 $ROOT/tools/syn_side_trace.py 20
prof train_atomic python3 $ROOT/tools/train_trace.py 50
VD_DETERMINISTIC=1 prof train_deterministic python3 $ROOT/tools/train_trace.py 50
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 280 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/tools/run_real_side.py 4 > $OUT/pmc_$c.log 2>&1
  cp $(ls $OUT/pmc_$c/*/*counter_collection.csv | head -1) $OUT/pmc_${c}_counter_collection.csv
done
cd $ROOT
tools/trace_step.sh prof_step > /dev/null 2>&1; cp gpurun_out/prof_step_timeline.txt $OUT/step_timeline.txt 2>/dev/null
python3 tools/pmc_real_side.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv 3200 $OUT/pmc_traffic.json > /dev/null
python3 tools/aux_kernel_gbps.py $OUT/aux_kernel_stats.csv $OUT/aux_kernels.json > /dev/null
rm -rf $OUT/bench $OUT/s2d $OUT/dc $OUT/mtt $OUT/aux $OUT/syn_side $OUT/train_atomic $OUT/train_deterministic $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
bash tools/pmc_sq.sh > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq $OUT/pmc_sq_summary.json > /dev/null 2>&1
cp gpurun_out/pmc_sq/pass*.csv $OUT/ 2>/dev/null
ls -la $OUT
