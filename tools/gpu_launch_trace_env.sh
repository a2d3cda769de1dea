#!/bin/bash
# like gpu_launch_trace.sh, all kernels of the last step with start offsets (several streams): <cfg> <sites> [ENV=..]
CFG=$1; N=$2; shift; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do export $V; done
rm -rf $R/gpurun_out/ltrace_env
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ltrace_env -- python3 $R/bench.py --config $CFG --sites $N --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ltrace_env.log 2>&1
f=$(ls -t $R/gpurun_out/ltrace_env/*/*_kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0] for r in rows]
ends = [i for i, n in enumerate(names) if 'edge_lnl' in n]
# last step = after the 4th-last .. group of lnl kernels; print the last 40 kernels
t0 = int(rows[-60]['Start_Timestamp']) if len(rows) > 60 else int(rows[0]['Start_Timestamp'])
for r in rows[-60:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"  q{r['Queue_Id']:>2s} {r['Kernel_Name'].split('(')[0][:34]:34s} grid {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>3s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:9.1f}")
PY
