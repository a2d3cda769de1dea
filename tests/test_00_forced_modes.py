"""The whole `-m gpu` suite once more under two engine-wide switches, each in a child process of its own (started
before this process has touched the GPU: the file sorts first):

  PLLHIP_SITE_REPEATS=2  every partition is treated as if it had been created with PLL_ATTRIB_SITE_REPEATS -- the
                         reference's test harness runs every program with and without the attribute and compares the
                         text (test/runtest.py:45-51, test/src/common.c:31);
  PLLHIP_TRANSIENT=1     every resident schedule runs evaluate-only (include/pllhip.h, pllhip_set_transient): vectors
                         inside operation chains are not stored and come back on demand.

Every oracle comparison, golden value and bit-for-bit check of the suite has to hold unchanged; the few asserts that
count launches or class operations are switched off through tests/common.py (FORCED_*).

A third child runs the tests of the Newton-Raphson loop over several partitions in its one-launch-per-partition form
(PLLHIP_NEWTON_ONE_LAUNCH=0) with GPU_MAX_HW_QUEUES=16: that form needs a hardware queue per partition stream, the library
leaves the runtime's four alone (pll_core.hip, newton_multi_enabled), and the variable has to be there before the process's
first HIP call.  (The one-launch form of the 4-, 20- and 33 .. 64-state families runs in the suite itself.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,env", [("site_repeats", {"PLLHIP_SITE_REPEATS": "2"}), ("transient", {"PLLHIP_TRANSIENT": "1"})])
def test_gpu_suite_under(name, env):
    if os.environ.get("PLLHIP_FORCED_CHILD"):
        pytest.skip("already inside a forced-mode run")
    cmd = [sys.executable, "-m", "pytest", "tests", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           "--ignore=tests/test_00_forced_modes.py"]
    r = subprocess.run(cmd, cwd=ROOT, env={**os.environ, **env, "PLLHIP_FORCED_CHILD": "1"}, capture_output=True,
                       text=True, timeout=1500)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"forced_{name}.log"), "w") as f:
            f.write(r.stdout + "\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout


def test_newton_loop_over_partitions_with_a_queue_per_stream():
    if os.environ.get("PLLHIP_FORCED_CHILD"):
        pytest.skip("already inside a forced-mode run")
    cmd = [sys.executable, "-m", "pytest", "tests/test_eval_driver.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           "-k", "several_partitions or newton_over_partitions"]
    r = subprocess.run(cmd, cwd=ROOT, env={**os.environ, "GPU_MAX_HW_QUEUES": "16", "PLLHIP_NEWTON_ONE_LAUNCH": "0", "PLLHIP_FORCED_CHILD": "1"},
                       capture_output=True, text=True, timeout=600)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "forced_newton_multi.log"), "w") as f:
            f.write(r.stdout + "\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1]
