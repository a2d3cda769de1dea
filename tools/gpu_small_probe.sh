#!/bin/bash
# small-launch changes: parity subset, SPR round at 61 states x 25 k sites, branch-length pass at 125 k protein sites
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_eval_driver.py tests/test_dropin_modules.py tests/test_mixture_models.py -x -q -m gpu > gpurun_out/small_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/small_tests.log
python tools/gpu_workloads.py spr25 blo125 > gpurun_out/wl_small.json 2> gpurun_out/wl_small.err
grep -h "s_per_round\|lnl_after\|us_per_scan\|s_per_smoothing" gpurun_out/wl_small.json
python tools/gpu_microbench.py 61 25000 30 2>&1 | tail -6
python tools/gpu_microbench.py 20 125000 30 2>&1 | tail -6
python bench.py --config c5 --no-also --no-cpu-baseline --steps 30 --warmup 5 2> gpurun_out/c5_small.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['ms_per_step'])"
python bench.py --config c5 --sites 25000 --no-also --no-cpu-baseline --steps 50 --warmup 5 2> gpurun_out/c5s_small.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 25k', d['ms_per_step'])"
