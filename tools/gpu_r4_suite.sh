#!/bin/bash
# the whole -m gpu suite as the driver runs it (incl. the forced-mode children of tests/test_00_forced_modes.py), with durations
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 > gpurun_out/r4_suite_full.log 2>&1
rc=$?; tail -45 gpurun_out/r4_suite_full.log; echo "rc $rc"
for f in gpurun_out/forced_*.log; do echo "== $f"; grep -E "^FAILED|^ERROR| passed| failed" $f | tail -12; done
