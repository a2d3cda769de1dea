#!/bin/bash
# Runs on the GPU box (gpurun): default bench, rocprofv3 kernel trace of the same
# command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for HBM traffic.
# Raw output lands in gpurun_out/; tools/summarize_profiles.py condenses it into
# profiles/.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
CFG=${1:-c3}
python bench.py --config $CFG --pmc on > gpurun_out/bench_$CFG.json 2> gpurun_out/bench_$CFG.err
tail -c 2500 gpurun_out/bench_$CFG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$CFG -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$CFG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$CFG -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_$CFG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$CFG -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write_$CFG.log 2>&1
ls -R $R/gpurun_out | head -40
