#!/bin/bash
# many partitions / the branch-length pass over C4's partitions across library revisions on ONE box
# (tools/ab/<sha>/ = bench.py + tools + library of that revision); GPU_MAX_HW_QUEUES left to the library, then forced to 4
mkdir -p gpurun_out
R=$PWD
for rep in 1 2; do for s in 8ca04ca 425630e HEAD; do
  if [ $s = HEAD ]; then d=$R; else d=$R/tools/ab/$s; fi
  for spec in "4 64 10000" "20 32 10000"; do
    (cd $d && timeout -k 10 200 python tools/gpu_many_partitions.py $spec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s', '$spec', round(d['ms_per_evaluation'],3), 'ms')") || echo "$s $spec failed"
  done
  if [ $s != 8ca04ca ]; then
    (cd $d && timeout -k 10 200 python tools/gpu_workloads.py blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['BLO_c4_125000']; print('$s', 'blo_c4_125', round(d['us_per_derivative_call_incl_everything'],2), 'us per iterate', d['device_newton'])") || echo "$s blo failed"
  fi
done; done
echo "-- GPU_MAX_HW_QUEUES=4"
for s in HEAD; do d=$R
  for spec in "4 64 10000" "20 32 10000"; do
    (cd $d && GPU_MAX_HW_QUEUES=4 timeout -k 10 200 python tools/gpu_many_partitions.py $spec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s q4', '$spec', round(d['ms_per_evaluation'],3), 'ms')") || echo "$s $spec failed"
  done
  (cd $d && GPU_MAX_HW_QUEUES=4 PLLHIP_EVAL_DEVICE_NEWTON=0 timeout -k 10 200 python tools/gpu_workloads.py blo_c4_125 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['BLO_c4_125000']; print('$s q4 host loop', 'blo_c4_125', round(d['us_per_derivative_call_incl_everything'],2), 'us per iterate', d['device_newton'])") || echo "$s blo failed"
done
