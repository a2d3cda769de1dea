#!/bin/bash
# round 4, final numbers, call A: default bench line (in-run PMC passes + also legs) + rocprofv3 kernel stats + FETCH / WRITE
# passes for c3 and c2
mkdir -p gpurun_out
bash tools/gpu_profile.sh c3 > gpurun_out/profile_c3.txt 2>&1; echo "c3 done"; tail -c 1500 gpurun_out/bench_c3.json
bash tools/gpu_profile.sh c2 > gpurun_out/profile_c2.txt 2>&1; echo "c2 done"; tail -c 1500 gpurun_out/bench_c2.json
