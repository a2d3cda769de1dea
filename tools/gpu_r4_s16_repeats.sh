#!/bin/bash
# site repeats in the 2 .. 32-state family: its tests, then the alphabet / parity files with the attribute forced on
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_site_repeats.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4_s16rep_a.log 2>&1; rc=$?
tail -15 gpurun_out/r4_s16rep_a.log; [ $rc = 0 ] || exit 1
PLLHIP_SITE_REPEATS=2 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_results.py tests/test_eval_driver.py -q -m gpu -p no:cacheprovider > gpurun_out/r4_s16rep_b.log 2>&1; rc=$?
tail -15 gpurun_out/r4_s16rep_b.log; [ $rc = 0 ] || exit 1
