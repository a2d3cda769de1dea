#!/bin/bash
# tools/gpu_probe.sh <states> <sites>: derivative-scan kernel times, both reduction finishes
R=$GRAFT_REPO_ROOT
S=${1:-20}; N=${2:-125000}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for F in 1 0; do
  export PLLHIP_FUSED_FINISH=$F
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/probe_${S}_${N}_f$F -- python3 $R/tools/gpu_deriv_probe.py $S $N 1 3 > $R/gpurun_out/probe_${S}_${N}_f$F.log 2>&1
  echo "== fused=$F"; grep "us per call" $R/gpurun_out/probe_${S}_${N}_f$F.log
  python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/probe_${S}_${N}_f$F/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'deriv' in r['Name'] or 'final' in r['Name'] or 'edge' in r['Name']:
            print(r['Name'].split('(')[0][:50].ljust(50), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
