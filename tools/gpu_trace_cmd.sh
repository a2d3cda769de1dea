#!/bin/bash
# tools/gpu_trace_cmd.sh <label> <program> [args...]: rocprofv3 --kernel-trace --stats of one command; the per-kernel
# summary goes to gpurun_out/trace_<label>_kernel_stats.csv (first 25 lines printed)
cd "$(dirname "$0")/.."
ROOT="$PWD"
label=$1; shift
export TMPDIR=/tmp
d=$ROOT/gpurun_out/trace_$label
rm -rf "$d"; mkdir -p "$d"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o t --output-format csv -- "$@" > "$d/stdout.txt" 2> "$d/stderr.txt") || { tail -5 "$d/stderr.txt"; exit 1; }
f=$(find "$d" -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/trace_${label}_kernel_stats.csv"
head -25 "$f" | cut -c1-180
tail -2 "$d/stdout.txt" | cut -c1-300
rm -rf "$d"
