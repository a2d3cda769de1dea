#!/bin/bash
# staging-loop change: parity subset, then single-operation microbenchmarks at small sizes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_site_repeats.py tests/test_mixture_models.py -x -q -m gpu > gpurun_out/stage_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/stage_tests.log
for cfg in "61 25000 50" "61 200000 20" "20 40000 50" "20 125000 30" "4 125000 50" "16 60000 50"; do
  set -- $cfg
  python tools/gpu_microbench.py $1 $2 $3 > gpurun_out/stage_mb_$1_$2.txt 2>&1
  PLLHIP_S61_WAVES8=4 python tools/gpu_microbench.py $1 $2 $3 > gpurun_out/stage_mb_$1_$2_w8.txt 2>&1
  echo "== $cfg"; cat gpurun_out/stage_mb_$1_$2.txt
done
echo "== 61 25000 waves8"; head -6 gpurun_out/stage_mb_61_25000_w8.txt
