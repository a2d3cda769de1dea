#!/bin/bash
# rocprofv3 kernel-trace summary of an arbitrary python tool: tools/gpu_trace.sh <name> <script> [args...]
NAME=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$NAME -- python3 "$@" > $R/gpurun_out/trace_$NAME.log 2>&1
f=$(ls -t $R/gpurun_out/trace_$NAME/*/*_kernel_stats.csv | head -1)
cut -c1-60 --complement $f > /dev/null 2>&1
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'].split('(')[0][:44]:44s} calls={int(r['Calls']):7d} total_ms={int(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
