#!/bin/bash
# schedules of a few operations in the kernel arguments (no copy in front of the launch): tests, then the branch-length passes,
# W3 and the default C2 / C3 lines (their kernels read their schedules through the same code)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_eval_driver.py tests/test_site_repeats.py tests/test_transient.py tests/test_partition_batch.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4_16_a.log 2>&1; rc=$?
tail -4 gpurun_out/r4_16_a.log; [ $rc = 0 ] || exit 1
for inl in 1 0; do
  PLLHIP_PLAN_INLINE=$inl timeout -k 10 300 python tools/gpu_workloads.py blo125 blo_c2 blo_c4_125 w3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('inline $inl', {k: round(v.get('us_per_derivative_call_incl_everything', v.get('us_per_iteration', 0)),2) for k,v in d.items()})" || echo "failed"
done
for cfg in c2 c3; do python bench.py --config $cfg --no-also --no-cpu-baseline --pmc off --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3))"; done
