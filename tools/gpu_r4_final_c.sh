#!/bin/bash
# round 4, final numbers, call C: strong-scaling slices of every configuration, secondary workloads
mkdir -p gpurun_out
rm -f gpurun_out/slices.jsonl
bash tools/gpu_slices.sh c3 1000000; echo "c3 slices done"
bash tools/gpu_slices.sh c2 1000000; echo "c2 slices done"
bash tools/gpu_slices.sh c4 1000000; echo "c4 slices done"
bash tools/gpu_slices.sh c5 200000; echo "c5 slices done"
timeout -k 10 700 python tools/gpu_workloads.py w2 w3 blo blo_c2 blo_c4 spr alphabets alphabets32 alphabets_repeats > gpurun_out/wl_r4.json 2> gpurun_out/wl_r4.err; echo "wl rc=$?"
PLLHIP_EVAL_DEVICE_NEWTON=0 timeout -k 10 200 python tools/gpu_workloads.py blo125 blo_c4_125 > gpurun_out/wl_r4_hostnewton.json 2>> gpurun_out/wl_r4.err; echo "wl host newton rc=$?"
: > gpurun_out/manypart_r4.jsonl
for spec in "20 32 10000" "20 1 320000" "20 8 40000" "4 64 10000" "4 1 640000" "4 16 60000"; do
  python tools/gpu_many_partitions.py $spec >> gpurun_out/manypart_r4.jsonl 2>>gpurun_out/wl_r4.err
done
python tools/gpu_newton_probe.py > gpurun_out/newton_probe_r4.txt 2>&1
echo "part C done"
