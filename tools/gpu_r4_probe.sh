#!/bin/bash
echo "== BLO workload, device loop, debug"; PLLHIP_NEWTON_DEBUG=1 PLLHIP_NEWTON_SPIN_LIMIT=300000 timeout -k 5 200 python tools/gpu_workloads.py blo_c4_125 2>&1 | grep -E "newton stuck|s_per_smoothing|us_per_deriv|lnl_after"
echo "== host loop"; PLLHIP_EVAL_DEVICE_NEWTON=0 timeout -k 5 200 python tools/gpu_workloads.py blo_c4_125 2>&1 | grep -E "s_per_smoothing|us_per_deriv|lnl_after"
echo "== probe"; timeout -k 5 120 python tools/gpu_newton_multi_probe.py
timeout -k 10 300 python -m pytest tests/test_eval_driver.py tests/test_gpu_results.py -q -x -p no:cacheprovider -k "newton or deferred or spread or speculative" 2>&1 | tail -3
