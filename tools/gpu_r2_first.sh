#!/bin/bash
# round 2, first GPU pass: new tests, then A/B of the reduction finish on the NR workload
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_results.py -x -q > gpurun_out/t_results.log 2>&1; echo "results rc=$?"; tail -5 gpurun_out/t_results.log
timeout -k 10 300 python tools/gpu_workloads.py blo125 w2 > gpurun_out/wl_fused1.json 2> gpurun_out/wl_fused1.err; echo "wl1 rc=$?"
PLLHIP_FUSED_FINISH=0 timeout -k 10 300 python tools/gpu_workloads.py blo125 w2 > gpurun_out/wl_fused0.json 2> gpurun_out/wl_fused0.err; echo "wl0 rc=$?"
PLLHIP_WORKLOAD_ATTACH=0 timeout -k 10 300 python tools/gpu_workloads.py blo125 > gpurun_out/wl_noattach.json 2> gpurun_out/wl_noattach.err; echo "wl-noattach rc=$?"
cat gpurun_out/wl_fused1.json gpurun_out/wl_fused0.json gpurun_out/wl_noattach.json
