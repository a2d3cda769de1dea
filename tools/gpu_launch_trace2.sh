#!/bin/bash
# per-launch durations of the last step of any bench.py command line: gpu_launch_trace2.sh <label> <bench args...>
R=$GRAFT_REPO_ROOT
label=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ltrace_$label
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ltrace_$label -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ltrace_$label.log 2>&1
f=$(ls -t $R/gpurun_out/ltrace_$label/*/*_kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0] for r in rows]
ends = [i for i, n in enumerate(names) if 'edge_lnl' in n]
a, b = ends[-2] + 1, ends[-1] + 1
prev_end = int(rows[a - 1]['End_Timestamp'])
tot = gap = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"  {r['Kernel_Name'].split('(')[0][:36]:36s} grid {r['Grid_Size_X']:>7s}x{r['Grid_Size_Y']:>4s} dur {(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f} us")
    tot += e - s; gap += s - prev_end; prev_end = e
print(f"  kernels {tot / 1e6:.3f} ms  gaps {gap / 1e6:.3f} ms")
PY
