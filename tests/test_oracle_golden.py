"""The CPU oracle against the reference's own golden vectors.

Two layers of pinning:
 1. `test_reference_programs_*`: where /root/reference is present (the build
    container), the reference's OWN test programs blopt-minimal.c / blopt-5states.c
    and its OWN optimiser sources (src/optimize/*.c) are compiled where they lie,
    against include/pll.h, linked with the oracle, run, and their stdout is
    compared byte for byte with the reference's golden .out files -- the same
    check test/runtest.py performs.  Objects go to a temp dir, nothing is copied.
 2. `test_fixture_*`: the same numbers through the C ABI from Python with the
    committed fixtures (tests/golden/blopt_fixtures.json), runnable anywhere.
"""
import glob
import os
import subprocess

import pytest

import common
from conftest import ROOT, ORACLE_LIB

REF = "/root/reference"


def _build_and_run(tmp_path, name, args):
    exe = tmp_path / name
    src = [f"{REF}/test/src/optimize/{name}.c", f"{REF}/test/src/common.c",
           f"{REF}/src/optimize/pll_optimize.c", f"{REF}/src/optimize/opt_algorithms.c",
           f"{REF}/src/pllmod_common.c"] + sorted(glob.glob(f"{REF}/src/optimize/lbfgsb/*.c"))
    inc = ["-I", f"{ROOT}/include", "-I", f"{REF}/src", "-I", f"{REF}/src/optimize",
           "-I", f"{REF}/src/tree", "-I", f"{REF}/test/src"]
    libdir = os.path.dirname(ORACLE_LIB)
    subprocess.run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-O2", "-w", *inc, "-o", str(exe), *src,
                    "-L", libdir, "-lpll_oracle", "-lm", f"-Wl,-rpath,{libdir}"], check=True)
    return subprocess.run([str(exe), *args], check=True, capture_output=True, text=True).stdout


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this box")
@pytest.mark.parametrize("name,args", [("blopt-minimal", []), ("blopt-minimal", ["avx"]),
                                       ("blopt-5states", []), ("blopt-5states", ["tv"])])
def test_reference_programs_reproduce_golden_output(oracle, tmp_path, name, args):
    out = _build_and_run(tmp_path, name, args)
    golden = open(os.path.join(common.GOLDEN, name + ".out")).read()
    assert out == golden


@pytest.mark.parametrize("name,coded", [("blopt-minimal", False), ("blopt-5states", False),
                                        ("blopt-5states", True)])
def test_fixture_through_c_abi(oracle, name, coded):
    expected, got = common.run_golden_case(oracle, name, coded=coded)
    common.check_golden(expected, got)


def test_gamma_rates_mean_mode(oracle):
    # Yang-1994 mean discretisation, alpha = 0.841, K = 4
    r = oracle.gamma_cats(0.841, 4)
    assert abs(r.mean() - 1.0) < 1e-7
    for got, exp in zip(r, [0.10424838, 0.42313162, 0.96769858, 2.50492142]):
        assert abs(got - exp) < 5e-7
