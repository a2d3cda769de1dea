#!/bin/bash
# round 2: remaining profiles in one call (C3 is taken by tools/gpu_profile.sh c3)
mkdir -p gpurun_out
bash tools/gpu_profile.sh c2 > gpurun_out/profile_c2.txt 2>&1; echo "c2 done"
bash tools/gpu_profile.sh c5 > gpurun_out/profile_c5.txt 2>&1; echo "c5 done"
python bench.py --config c4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"
python bench.py --config c5 --rate-scalers > gpurun_out/bench_c5_rs.json 2> gpurun_out/bench_c5_rs.err; echo "c5 rs rc=$?"
python bench.py --config c3 --rate-scalers --no-cpu-baseline > gpurun_out/bench_c3_rs.json 2> gpurun_out/bench_c3_rs.err; echo "c3 rs rc=$?"
timeout -k 10 400 python tools/gpu_workloads.py alphabets w2 w3 c4 > gpurun_out/wl_r2_a.json 2> gpurun_out/wl_r2_a.err; echo "wl rc=$?"
PLLHIP_NO_S16=1 timeout -k 10 300 python tools/gpu_workloads.py alphabets > gpurun_out/wl_r2_generic.json 2> gpurun_out/wl_r2_generic.err; echo "wl generic rc=$?"
