#!/bin/bash
# C3 / C2 default lines: the head library against the one with schedules of a few operations in the kernel arguments
# (tools/ab/inline), same box
mkdir -p gpurun_out
R=$PWD
for rep in 1 2 3; do for s in HEAD inline; do
  if [ $s = HEAD ]; then d=$R; else d=$R/tools/ab/$s; fi
  for cfg in c3 c2; do
    (cd $d && python bench.py --config $cfg --no-also --no-cpu-baseline --pmc off --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s', '$cfg', round(d['ms_per_step'],3), flush=True)") || echo "$s failed"
  done
done; done
