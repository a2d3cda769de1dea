#!/bin/bash
# site repeats in the 2 .. 32-state family and without pattern tips: their tests, then the whole suite (with the forced-mode children)
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_site_repeats.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4_s16rep_a.log 2>&1; rc=$?
tail -15 gpurun_out/r4_s16rep_a.log; [ $rc = 0 ] || exit 1
bash tools/gpu_r4_suite.sh
