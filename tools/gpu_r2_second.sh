#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_results.py -q > gpurun_out/t_results.log 2>&1; echo "results rc=$?"; tail -8 gpurun_out/t_results.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blo125 -- python3 $R/tools/gpu_workloads.py blo125 > $R/gpurun_out/prof_blo125.log 2>&1; echo "prof rc=$?"
cat $R/gpurun_out/prof_blo125/*/*_kernel_stats.csv | cut -c1-200
