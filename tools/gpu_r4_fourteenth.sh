#!/bin/bash
# the Newton loop over several partitions in ONE launch (k_newton_multi): its tests (both forms), then the C4 branch-length pass
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_eval_driver.py tests/test_gpu_results.py tests/test_00_forced_modes.py -q -m gpu -x -p no:cacheprovider -k "newton or driver or partitions" > gpurun_out/r4_14_a.log 2>&1; rc=$?
tail -6 gpurun_out/r4_14_a.log; [ $rc = 0 ] || exit 1
for rep in 1 2; do
  for dev in 1 0; do
    PLLHIP_EVAL_DEVICE_NEWTON=$dev timeout -k 10 200 python tools/gpu_workloads.py blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['BLO_c4_125000']; print('blo_c4_125 device loop $dev', round(d['us_per_derivative_call_incl_everything'],2), 'us per iterate', d['newton_iterations'], d['lnl_after'])" || echo "blo failed"
  done
  PLLHIP_EVAL_DEVICE_NEWTON=1 timeout -k 10 200 python tools/gpu_workloads.py blo_c4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v['us_per_derivative_call_incl_everything'],2) for k,v in d.items()})" || echo "blo failed"
done
