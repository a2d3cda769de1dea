#!/bin/bash
# the Newton loop over several partitions on streams with hardware queues of their own (stream priorities), the runtime's
# four queues left alone: its tests, then many partitions / the C4 branch-length pass / C4
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_eval_driver.py tests/test_gpu_results.py -q -m gpu -x -p no:cacheprovider -k "newton or driver or partitions" > gpurun_out/r4_13_a.log 2>&1; rc=$?
tail -6 gpurun_out/r4_13_a.log; [ $rc = 0 ] || exit 1
for rep in 1 2; do
  for spec in "4 64 10000" "20 32 10000" "4 16 60000"; do
    timeout -k 10 200 python tools/gpu_many_partitions.py $spec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec', round(d['ms_per_evaluation'],3), 'ms')" || echo "$spec failed"
  done
  for dev in 1 0; do
    PLLHIP_EVAL_DEVICE_NEWTON=$dev timeout -k 10 200 python tools/gpu_workloads.py blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['BLO_c4_125000']; print('blo_c4_125 device loop $dev', round(d['us_per_derivative_call_incl_everything'],2), 'us per iterate', d['newton_iterations'], d['lnl_after'])" || echo "blo failed"
  done
  timeout -k 10 200 python bench.py --config c4 --no-also --no-cpu-baseline --pmc off --steps 20 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', round(d['ms_per_step'],3), 'ms/step')" || echo "c4 failed"
done
