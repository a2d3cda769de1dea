#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: tools/gpu_trace.sh <name> <script> [args...]
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$name -- python3 $R/"$@" > $R/gpurun_out/trace_$name.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/trace_$name/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k:45s} n={len(v):5d} avg={sum(v)/len(v)/1e3:9.1f} us  med={v[len(v)//2]/1e3:9.1f}  min={v[0]/1e3:8.1f} max={v[-1]/1e3:9.1f}")
PY
