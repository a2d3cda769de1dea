#!/bin/bash
# round 3, final numbers, call A: default bench line + rocprofv3 kernel stats + FETCH / WRITE passes for c3 and c2
mkdir -p gpurun_out
bash tools/gpu_profile.sh c3 > gpurun_out/profile_c3.txt 2>&1; echo "c3 done"
bash tools/gpu_profile.sh c2 > gpurun_out/profile_c2.txt 2>&1; echo "c2 done"
