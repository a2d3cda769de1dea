#!/bin/bash
# hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES) against the many-stream workloads and the co-resident Newton loops
mkdir -p gpurun_out
for q in 4 6 8 12 16; do
  for spec in "4 64 10000" "20 32 10000" "4 16 60000"; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python tools/gpu_many_partitions.py $spec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('q$q', '$spec', round(d['ms_per_evaluation'],3), 'ms')" || echo "q$q $spec failed"
  done
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python tools/gpu_workloads.py blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['BLO_c4_125000']; print('q$q', 'blo_c4_125', round(d['us_per_derivative_call_incl_everything'],2), 'us per iterate, device loop still on:', d['device_newton'])" || echo "q$q blo failed"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --config c4 --no-also --no-cpu-baseline --pmc off --steps 20 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('q$q', 'c4', round(d['ms_per_step'],3), 'ms/step')" || echo "q$q c4 failed"
done
