#!/bin/bash
# round 3, part 3: secondary workloads
mkdir -p gpurun_out
timeout -k 10 700 python tools/gpu_workloads.py w2 w3 blo blo_c2 spr alphabets32 > gpurun_out/wl_r3.json 2> gpurun_out/wl_r3.err; echo "wl rc=$?"
PLLHIP_EVAL_DEVICE_NEWTON=0 timeout -k 10 200 python tools/gpu_workloads.py blo125 > gpurun_out/wl_r3_hostnewton.json 2>> gpurun_out/wl_r3.err; echo "wl host newton rc=$?"
: > gpurun_out/manypart_r3.jsonl
for spec in "20 32 10000" "20 1 320000" "20 8 40000" "4 64 10000" "4 1 640000" "4 16 60000"; do
  python tools/gpu_many_partitions.py $spec >> gpurun_out/manypart_r3.jsonl 2>>gpurun_out/wl_r3.err
done
PLLHIP_BATCH=0 PLLHIP_EVAL_DEFERRED=0 python tools/gpu_many_partitions.py 20 32 10000 | sed 's/^{/{"variant": "per-partition launches, one wait per partition (round 2)", /' >> gpurun_out/manypart_r3.jsonl
python tools/gpu_newton_probe.py > gpurun_out/newton_probe_r3.txt 2>&1
echo "part3 done"
