#!/bin/bash
# the Newton control block initialised from kernel arguments (no copy per loop): the tests of the loops (incl. the stall
# injection and the one-launch-per-partition form), then the branch-length passes
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_eval_driver.py tests/test_gpu_results.py tests/test_00_forced_modes.py -q -m gpu -x -p no:cacheprovider -k "newton or driver or partitions" > gpurun_out/r4_15_a.log 2>&1; rc=$?
tail -5 gpurun_out/r4_15_a.log; [ $rc = 0 ] || exit 1
for rep in 1 2; do
  timeout -k 10 300 python tools/gpu_workloads.py blo125 blo_c2 blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v['us_per_derivative_call_incl_everything'],2) for k,v in d.items()})" || echo "blo failed"
done
