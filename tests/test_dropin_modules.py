"""Drop-in check of the boundary at the level pll-modules' clients use it.

Where /root/reference is present, the reference's UNMODIFIED upper layers --
src/tree/treeinfo.c, src/optimize/pll_optimize.c + opt_algorithms.c + lbfgsb,
src/algorithm/algo_search.c / pllmod_algorithm.c / algo_callback.c and the tree
utilities they pull in -- are compiled where they lie against include/pll.h and
linked with this repository's library (the CPU oracle build of the same
interface, since this container has no GPU).  A small client
(tests/dropin/treeinfo_driver.c) then runs a two-partition analysis: full and
incremental likelihood, multi-partition Newton-Raphson branch-length
optimisation, and one pllmod_algo_spr_round.  The reference's own internal
assertions (e.g. incremental lnL == full lnL after the SPR round,
src/algorithm/algo_search.c:1453-1457) are active.

This is a test of the BOUNDARY (names, struct fields, semantics of ~35 libpll
functions incl. the SPR/NNI primitives and the partial-traversal callbacks), not
an oracle: the arithmetic underneath is this repository's.
"""
import glob
import os
import re
import subprocess

import pytest

from conftest import ROOT, ORACLE_LIB

REF = "/root/reference"
MODULES = ["src/pllmod_common.c", "src/tree/treeinfo.c", "src/tree/utree_operations.c",
           "src/tree/rtree_operations.c", "src/tree/pll_tree.c", "src/tree/utree_constraint.c",
           "src/tree/utree_distances.c", "src/tree/tree_hashtable.c", "src/tree/consensus.c",
           "src/util/models.c", "src/util/models_dna.c", "src/util/models_gt.c", "src/util/models_mult.c",
           "src/optimize/pll_optimize.c", "src/optimize/opt_algorithms.c",
           "src/algorithm/algo_search.c", "src/algorithm/algo_callback.c",
           "src/algorithm/pllmod_algorithm.c"]


@pytest.fixture(scope="module")
def driver(oracle, tmp_path_factory):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this box")
    exe = tmp_path_factory.mktemp("dropin") / "ti_driver"
    src = [os.path.join(ROOT, "tests", "dropin", "treeinfo_driver.c")]
    src += [os.path.join(REF, m) for m in MODULES] + sorted(glob.glob(f"{REF}/src/optimize/lbfgsb/*.c"))
    inc = sum((["-I", d] for d in [f"{ROOT}/include", f"{REF}/src", f"{REF}/src/optimize", f"{REF}/src/tree",
                                   f"{REF}/src/algorithm", f"{REF}/src/util"]), [])
    libdir = os.path.dirname(ORACLE_LIB)
    subprocess.run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-O2", "-w", *inc, "-o", str(exe), *src,
                    "-L", libdir, "-lpll_oracle", "-lm", f"-Wl,-rpath,{libdir}"], check=True)
    return str(exe)


@pytest.mark.parametrize("args", [[], ["tv"]])
def test_reference_treeinfo_blo_and_spr_round_run_on_this_library(driver, args):
    out = subprocess.run([driver, *args], check=True, capture_output=True, text=True, timeout=600).stdout
    v = {k.strip(): float(x) for k, x in re.findall(r"^(.*?):\s+(-?[0-9.]+)$", out, re.M)}
    assert set(v) == {"direct lnL", "driver full lnL", "driver incremental", "driver after BLO",
                      "full lnL", "incremental lnL", "after BLO", "after SPR round", "full recomputation"}
    # this repository's evaluation driver (include/pllhip_eval.h) against the
    # reference's treeinfo + Newton-Raphson optimiser on the same data and settings
    assert abs(v["driver full lnL"] - v["full lnL"]) < 1e-6
    assert abs(v["driver incremental"] - v["full lnL"]) < 1e-6
    assert abs(v["driver after BLO"] - v["after BLO"]) < 1e-4
    assert abs(v["direct lnL"] - v["full lnL"]) < 1e-6          # raw pll_* calls == treeinfo
    assert abs(v["incremental lnL"] - v["full lnL"]) < 1e-6
    assert v["after BLO"] > v["full lnL"] + 1.0
    assert v["after SPR round"] >= v["after BLO"] - 1e-6
    assert abs(v["full recomputation"] - v["after SPR round"]) < 1e-6


def test_tip_pattern_mode_gives_the_same_numbers(driver):
    a = subprocess.run([driver], check=True, capture_output=True, text=True).stdout
    b = subprocess.run([driver, "tv"], check=True, capture_output=True, text=True).stdout
    assert a == b       # the reference's own cross-backend criterion (test/runtest.py:45-51)
