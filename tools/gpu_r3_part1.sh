#!/bin/bash
# round 3, part 1: the default bench line + rocprofv3 kernel stats + FETCH / WRITE passes for c3, c2, c5; bench lines of c4
mkdir -p gpurun_out
sed -i 's/--steps 5 --warmup 2 --no-cpu-baseline >/--steps 5 --warmup 2 --no-cpu-baseline >/' tools/gpu_profile.sh
bash tools/gpu_profile.sh c3 > gpurun_out/profile_c3.txt 2>&1; echo "c3 done"
bash tools/gpu_profile.sh c2 > gpurun_out/profile_c2.txt 2>&1; echo "c2 done"
bash tools/gpu_profile.sh c5 > gpurun_out/profile_c5.txt 2>&1; echo "c5 done"
python bench.py --config c4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"
python bench.py --site-repeats --no-cpu-baseline > gpurun_out/bench_c3_repeats.json 2> gpurun_out/bench_c3_repeats.err; echo "c3 repeats rc=$?"
python bench.py --site-repeats --data simulated --no-cpu-baseline > gpurun_out/bench_c3_repeats_sim.json 2>> gpurun_out/bench_c3_repeats.err; echo "c3 repeats sim rc=$?"
