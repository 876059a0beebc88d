#!/bin/bash
python -m pytest tests/test_gpu_parity_late.py tests/test_gpu_dither.py tests/test_gpu_collectives.py tests/test_gpu_e2e.py -q -s -k "full_size or last_level or s2d_dc_mtt or e2e or oracle_loop or vd_comm" 2>&1 | grep -v "^$" > gpurun_out/r03_t4.log
grep -E "FAILED|passed|failed|DM 12 steps|evaluate_synset \(|shipped  vs|clean entries|entries with|^E  " gpurun_out/r03_t4.log | cut -c1-400
python tools/eval_epochs_probe.py
