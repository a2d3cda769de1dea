#!/bin/bash
# round 4, final numbers, call B: profile of c5 (in-run PMC); bench lines of c4, of the evaluate-only and the site-repeats
# variants on random / simulated data
mkdir -p gpurun_out
bash tools/gpu_profile.sh c5 > gpurun_out/profile_c5.txt 2>&1; echo "c5 done"
python bench.py --config c4 --pmc on > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"
for cfg in c2 c3 c4; do
  for data in random simulated; do
    python bench.py --config $cfg --site-repeats --data $data --no-also --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/bench_${cfg}_repeats_${data}.json 2> gpurun_out/bench_${cfg}_repeats_${data}.err; echo "$cfg repeats $data rc=$?"
  done
  python bench.py --config $cfg --data simulated --no-also --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/bench_${cfg}_simulated.json 2> gpurun_out/bench_${cfg}_simulated.err; echo "$cfg simulated (attribute off) rc=$?"
  python bench.py --config $cfg --transient --no-also --no-cpu-baseline --pmc on --steps 20 --warmup 4 > gpurun_out/bench_${cfg}_transient.json 2> gpurun_out/bench_${cfg}_transient.err; echo "$cfg evaluate-only rc=$?"
done
