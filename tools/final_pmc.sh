#!/bin/bash
# re-measure the stamped PMC traffic file at the current sources + one default bench line
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_embed.py tests/test_gpu_cdriver.py -m gpu -x -q 2>&1 | tail -2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  timeout -k 5 280 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/tools/run_real_side.py 4 > $OUT/pmc_$c.log 2>&1
  cp $(ls $OUT/pmc_$c/*/*counter_collection.csv | head -1) $OUT/pmc_${c}_counter_collection.csv
  rm -rf $OUT/pmc_$c
done
cd $ROOT
python3 tools/pmc_real_side.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv 3200 $OUT/pmc_traffic.json > /dev/null
cp $OUT/pmc_traffic.json profiles/r06_pmc_traffic.json
python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_1gpu_final.json
python3 -c "
import json; b=json.load(open('$OUT/bench_1gpu_final.json')); r=b['roofline']; print(b['value'], b['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'][:90])"
