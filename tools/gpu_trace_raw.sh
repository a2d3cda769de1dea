#!/bin/bash
# tools/gpu_trace_raw.sh <label> <program> [args...]: rocprofv3 --kernel-trace of one command; per kernel the
# count / min / median / max duration and the median gap to the previous dispatch (us).  TRACE_WINDOW=first,count also
# lists that run of dispatches one by one (start relative to the first of them, duration, queue) in traceraw_<label>_window.txt
cd "$(dirname "$0")/.."
ROOT="$PWD"
label=$1; shift
export TMPDIR=/tmp
d=$ROOT/gpurun_out/traceraw_$label
rm -rf "$d"; mkdir -p "$d"
(cd /tmp && rocprofv3 --kernel-trace -d "$d" -o t --output-format csv -- "$@" > "$d/stdout.txt" 2> "$d/stderr.txt") || { tail -5 "$d/stderr.txt"; exit 1; }
f=$(find "$d" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$ROOT/gpurun_out/traceraw_${label}_window.txt" > "$ROOT/gpurun_out/traceraw_${label}.txt" <<'PY'
import csv, os, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if os.environ.get("TRACE_WINDOW"):
    a, n = (int(v) for v in os.environ["TRACE_WINDOW"].split(","))
    a = a if a >= 0 else len(rows) + a
    w = rows[a:a + n]
    with open(sys.argv[2], "w") as out:
        t0 = int(w[0]["Start_Timestamp"])
        for r in w:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            out.write(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3s}  grid {r.get('Grid_Size_X') or r.get('Grid_Size', '?'):>7s}  {r['Kernel_Name'].split('(')[0][-70:]}\n")
by = {}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    grid = (r.get("Grid_Size_X") or r.get("Grid_Size", "?"), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""), r.get("VGPR_Count", ""))
    key = (name, grid)
    by.setdefault(key, ([], []))
    by[key][0].append((e - s) / 1e3)
    if prev_end is not None: by[key][1].append((s - prev_end) / 1e3)
    prev_end = e
span = (max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = sum(sum(d) for d, g in by.values()) / 1e3
print(f"span {span:.1f} ms, kernels {busy:.1f} ms, idle before a kernel (gaps > 0, < 1 ms) {sum(x for d, g in by.values() for x in g if 0 < x < 1000) / 1e3:.1f} ms")
for (name, grid), (d, g) in sorted(by.items(), key=lambda kv: -sum(kv[1][0])):
    pos = [x for x in g if 0 < x < 1000]
    print(f"{name:60s} grid/wg/lds/scratch/vgpr={grid} n={len(d):5d} dur min {min(d):7.1f} med {st.median(d):7.1f} max {max(d):7.1f} sum {sum(d) / 1e3:7.1f} ms  gap med {st.median(g) if g else 0:6.1f} sum {sum(pos) / 1e3:6.1f} ms")
PY
head -40 "$ROOT/gpurun_out/traceraw_${label}.txt" | cut -c1-230
rm -rf "$d"
