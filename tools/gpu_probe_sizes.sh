#!/bin/bash
R=$GRAFT_REPO_ROOT
S=${1:-20}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for N in 2000 30000 60000 125000 250000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/psz_${S}_${N} -- python3 $R/tools/gpu_deriv_probe.py $S $N 1 > $R/gpurun_out/psz_${S}_${N}.log 2>&1
  echo "== N=$N"; grep "us per call" $R/gpurun_out/psz_${S}_${N}.log
  python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/psz_${S}_${N}/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'deriv' in r['Name'] or 'edge' in r['Name']:
            print(r['Name'].split('(')[0][:50].ljust(50), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
