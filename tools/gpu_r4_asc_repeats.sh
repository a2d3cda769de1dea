#!/bin/bash
# site repeats on ascertainment-bias partitions: the site-repeats tests, the ascertainment tests plain and with the attribute forced
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_site_repeats.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4_asc_a.log 2>&1; rc=$?
tail -8 gpurun_out/r4_asc_a.log; [ $rc = 0 ] || exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "ascertainment" -p no:cacheprovider > gpurun_out/r4_asc_b.log 2>&1; rc=$?
tail -4 gpurun_out/r4_asc_b.log; [ $rc = 0 ] || exit 1
PLLHIP_SITE_REPEATS=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_results.py -q -m gpu -p no:cacheprovider > gpurun_out/r4_asc_c.log 2>&1; rc=$?
tail -8 gpurun_out/r4_asc_c.log; [ $rc = 0 ] || exit 1
