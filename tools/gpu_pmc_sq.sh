#!/bin/bash
# SQ counters of the bench kernels (MFMA busy cycles, LDS bank conflicts, wave/wait cycles):
# tools/gpu_pmc_sq.sh <cfg>; raw output in gpurun_out/pmc_sq_<cfg>, condensed into profiles/ by
# tools/summarize_sq.py.  One --pmc pass (8 SQ slots), no tracing domains.
CFG=${1:-c3}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_$CFG -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq_$CFG.log 2>&1
tail -2 $R/gpurun_out/pmc_sq_$CFG.log
ls $R/gpurun_out/pmc_sq_$CFG/* | head
