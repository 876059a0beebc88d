cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_rank_proxy.txt; : > $OUT
for n in 1 2 4 8; do timeout 300 python tools/rank_proxy.py $n 2>/dev/null | grep "N=" >> $OUT; done
echo "== N = 8, VD_C8_POS=0" >> $OUT; VD_C8_POS=0 timeout 300 python tools/rank_proxy.py 8 2>/dev/null | grep "N=8" >> $OUT
echo "== N = 8, VD_PREP_MIN_CLIPS=0 (preparation stream on)" >> $OUT; VD_PREP_MIN_CLIPS=0 timeout 300 python tools/rank_proxy.py 8 2>/dev/null | grep "N=8" >> $OUT
cat $OUT
