#!/bin/bash
# gpurun_out/prof/* (written by tools/refresh_profiles.sh on the GPU box) -> profiles/r06_*
P=gpurun_out/prof
for f in bench_1gpu.json bench_1gpu_again.json bench_1gpu_under_rocprof.json bench_s2d.json bench_config1_shape.json bench_dc.json bench_mtt.json bench_mtt_bf16x3.json bench_dc_bf16x3.json mfma_rounding.txt \
         bench_8ranks_one_device.json mfma_peak.txt train_step.txt train_step_deterministic.txt aux_kernels.json \
         bench_kernel_stats.csv s2d_kernel_stats.csv dc_kernel_stats.csv mtt_kernel_stats.csv aux_kernel_stats.csv syn_side_kernel_stats.csv \
         train_atomic_kernel_stats.csv train_deterministic_kernel_stats.csv pmc_traffic.json pmc_sq_summary.json l0_kernel_ab.txt l0_phase_stamps.txt hal_bwd.txt rank_proxy.txt step_timeline.txt; do
  [ -s $P/$f ] && cp $P/$f profiles/r06_$f
done
[ -s $P/pmc_FETCH_SIZE_counter_collection.csv ] && cp $P/pmc_FETCH_SIZE_counter_collection.csv profiles/r06_pmc_fetch_size_counter_collection.csv
[ -s $P/pmc_WRITE_SIZE_counter_collection.csv ] && cp $P/pmc_WRITE_SIZE_counter_collection.csv profiles/r06_pmc_write_size_counter_collection.csv
for i in 1 2 3 4; do [ -s $P/pass$i.csv ] && cp $P/pass$i.csv profiles/r06_pmc_sq_pass$i.csv; done

ls -la profiles | grep r06_
