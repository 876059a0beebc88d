cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_bwd_x2sim_g.txt; : > $OUT
for m in g; do
  echo "== VD_BWD_X2_SIM='$m'" >> $OUT
  VD_BWD_X2_SIM=$m timeout 900 python -m pytest tests/test_gpu_parity_late.py -m gpu -q -s 2>&1 | grep -v "^$" | grep "vs oracle\|clean entries\|passed\|failed" >> $OUT
  cp gpurun_out/r04_parity.json gpurun_out/r04_parity_bwdx2sim_$m.json
done
cat $OUT
