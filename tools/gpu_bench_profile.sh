mkdir -p gpurun_out
set -x
R=$GRAFT_REPO_ROOT
python bench.py --config c2 --steps 5 --warmup 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -c 3000 gpurun_out/bench_c2.json
python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3_generic.json 2> gpurun_out/bench_c3.err
tail -c 2500 gpurun_out/bench_c3_generic.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_c2.log 2>&1
ls -R $R/gpurun_out/prof_c2 | head -20
