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


@pytest.fixture(scope="module")
def linkage_driver(oracle, tmp_path_factory):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this box")
    exe = tmp_path_factory.mktemp("dropin") / "linkage_driver"
    src = [os.path.join(ROOT, "tests", "dropin", "linkage_driver.c")]
    src += [os.path.join(REF, m) for m in MODULES] + sorted(glob.glob(f"{REF}/src/optimize/lbfgsb/*.c"))
    inc = sum((["-I", d] for d in [f"{ROOT}/include", f"{REF}/src", f"{REF}/src/optimize", f"{REF}/src/tree",
                                   f"{REF}/src/algorithm", f"{REF}/src/util"]), [])
    libdir = os.path.dirname(ORACLE_LIB)
    subprocess.run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-O2", "-w", *inc, "-o", str(exe), *src,
                    "-L", libdir, "-lpll_oracle", "-lm", f"-Wl,-rpath,{libdir}"], check=True)
    return str(exe)


@pytest.mark.parametrize("mode", ["scaled", "unlinked"])
def test_scaled_and_unlinked_branch_lengths_match_the_reference(linkage_driver, mode):
    """per-partition branch-length scalers / per-partition branch lengths: likelihood and
    Newton-Raphson optimisation of this repository's driver against the reference's treeinfo +
    pllmod_algo_opt_brlen_treeinfo (src/tree/treeinfo.c:176-183, 849-852;
    src/optimize/pll_optimize.c:1223-1287, 1395-1712) on identical data"""
    out = subprocess.run([linkage_driver, mode], check=True, capture_output=True, text=True, timeout=600).stdout
    v = {k.strip(): float(x) for k, x in re.findall(r"^(.*?):\s+(-?[0-9.]+)$", out, re.M)}
    assert abs(v["driver lnL"] - v["lnL"]) < 1e-6
    assert v["after BLO"] > v["lnL"] + 1.0
    assert abs(v["driver after BLO"] - v["after BLO"]) < 1e-5
    assert abs(v["driver re-eval"] - v["driver after BLO"]) < 1e-6 and abs(v["re-eval"] - v["after BLO"]) < 1e-6
    for p in range(3):
        assert abs(v[f"driver tree length {p}"] - v[f"tree length {p}"]) < 1e-5
        assert abs(v[f"driver branch 5 of {p}"] - v[f"branch 5 of {p}"]) < 1e-6
    if mode == "unlinked":      # the partitions really ended up with different lengths
        assert abs(v["tree length 0"] - v["tree length 2"]) > 0.1


def test_tip_pattern_mode_gives_the_same_numbers(driver):
    a = subprocess.run([driver], check=True, capture_output=True, text=True).stdout
    b = subprocess.run([driver, "tv"], check=True, capture_output=True, text=True).stdout
    assert a == b       # the reference's own cross-backend criterion (test/runtest.py:45-51)


# ---------------------------------------------------------------------------
# SPR round: the reference's pllmod_algo_spr_round against this repository's
# counterpart (pllhip_eval_spr_round) from the same starting tree
# ---------------------------------------------------------------------------
def _simulate(ntaxa, nsites, seed):
    """sequences evolved along a random tree (JC-like, two rate classes) and a
    scrambled starting tree over the same labels"""
    import numpy as np
    rng = np.random.default_rng(seed)

    def random_tree(order):
        items = [f"t{i}" for i in order]
        while len(items) > 3:
            i, j = sorted(rng.choice(len(items), 2, replace=False))
            b = items.pop(j)
            a = items.pop(i)
            items.append((a, b))
        return tuple(items)

    true = random_tree(range(ntaxa))
    seqs = {}

    def evolve(node, state, rate):
        if isinstance(node, str):
            seqs[node] = state
            return
        for child in node:
            t = rng.uniform(0.02, 0.25)
            p = 0.75 * (1.0 - np.exp(-4.0 / 3.0 * t * rate))
            change = rng.random(state.size) < p
            new = np.where(change, (state + rng.integers(1, 4, state.size)) % 4, state)
            evolve(child, new, rate)

    rate = np.where(rng.random(nsites) < 0.5, 0.3, 1.7)
    root = rng.integers(0, 4, nsites)
    for child in true:
        evolve(child, root if False else np.where(rng.random(nsites) < 0.05, (root + 1) % 4, root), rate)

    def newick(node):
        if isinstance(node, str):
            return f"{node}:{rng.uniform(0.03, 0.2):.6f}"
        return "(" + ",".join(newick(c) for c in node) + f"):{rng.uniform(0.03, 0.2):.6f}"

    start = random_tree(rng.permutation(ntaxa))
    nwk = "(" + ",".join(newick(c) for c in start) + ");"
    aln = "\n".join(f"{k} {''.join('acgt'[x] for x in v)}" for k, v in sorted(seqs.items())) + "\n"
    return nwk, aln


def _splits(nwk):
    """set of bipartitions (as frozensets of tip labels on the side without t0)"""
    labels = set(re.findall(r"t\d+", nwk))
    out, stack = set(), []
    for tok in re.findall(r"\(|\)|t\d+", nwk):
        if tok == "(":
            stack.append(set())
        elif tok == ")":
            s = stack.pop()
            if stack:
                stack[-1] |= s
            side = frozenset(s if "t0" not in s else labels - s)
            if 1 < len(side) < len(labels) - 1:
                out.add(side)
        else:
            if stack:
                stack[-1].add(tok)
    return out


def _brlens(nwk):
    return sorted(float(x) for x in re.findall(r":([0-9.eE+-]+)", nwk))


@pytest.fixture(scope="module")
def spr_driver(oracle, tmp_path_factory):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this box")
    exe = tmp_path_factory.mktemp("dropin_spr") / "spr_driver"
    src = [os.path.join(ROOT, "tests", "dropin", "spr_driver.c")]
    src += [os.path.join(REF, m) for m in MODULES] + sorted(glob.glob(f"{REF}/src/optimize/lbfgsb/*.c"))
    inc = sum((["-I", d] for d in [f"{ROOT}/include", f"{REF}/src", f"{REF}/src/optimize", f"{REF}/src/tree",
                                   f"{REF}/src/algorithm", f"{REF}/src/util"]), [])
    libdir = os.path.dirname(ORACLE_LIB)
    subprocess.run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-O2", "-w", *inc, "-o", str(exe), *src,
                    "-L", libdir, "-lpll_oracle", "-lm", f"-Wl,-rpath,{libdir}"], check=True)
    return str(exe)


@pytest.mark.parametrize("mode,ntaxa,radius,rounds,ntopol,seed,linkage", [
    ("fast", 14, 5, 2, 5, 1, "linked"),
    ("thorough", 12, 4, 1, 3, 2, "linked"),
    ("fast", 20, 7, 2, 8, 3, "linked"),
    ("thorough", 16, 3, 2, 4, 4, "linked"),
    # per-partition branch lengths (src/algorithm/algo_search.c:399-565, 639-641, 756, 811-813)
    ("fast", 14, 5, 2, 5, 5, "unlinked"),
    ("thorough", 12, 4, 2, 3, 6, "unlinked"),
    ("fast", 18, 6, 1, 6, 7, "scaled"),
])
def test_own_spr_round_makes_the_reference_s_moves(spr_driver, tmp_path, mode, ntaxa, radius, rounds,
                                                   ntopol, seed, linkage):
    nwk, aln = _simulate(ntaxa, 600, seed)
    (tmp_path / "start.nwk").write_text(nwk)
    (tmp_path / "aln.txt").write_text(aln)
    out = subprocess.run([spr_driver, str(tmp_path / "start.nwk"), str(tmp_path / "aln.txt"), mode,
                          str(radius), str(rounds), str(ntopol), linkage],
                         check=True, capture_output=True, text=True, timeout=900).stdout
    ref_tree = re.search(r"^ref tree: (.*)$", out, re.M).group(1)
    own_tree = re.search(r"^own tree: (.*)$", out, re.M).group(1)
    assert _splits(ref_tree) != _splits(nwk), "the round should have changed the starting topology"
    assert _splits(own_tree) == _splits(ref_tree)
    if linkage != "unlinked":            # (with per-partition lengths the tree's own lengths are not maintained)
        for a, b in zip(_brlens(own_tree), _brlens(ref_tree)):
            assert abs(a - b) < 1e-5
    for p in range(2):
        ref = float(re.search(rf"^ref plen {p}: (\S+)$", out, re.M).group(1))
        own = float(re.search(rf"^own plen {p}: (\S+)$", out, re.M).group(1))
        assert abs(ref - own) < 1e-4, (p, ref, own)
    for r in range(rounds):
        ref = float(re.search(rf"^ref round {r} lnL: (\S+)$", out, re.M).group(1))
        own = float(re.search(rf"^own round {r} lnL: (\S+)$", out, re.M).group(1))
        assert abs(ref - own) < 1e-5
        # the cutoff statistics accumulate over EVERY placement scored in the scan
        rc = re.search(rf"^ref round {r} cutoff: (\S+) (\S+) (\S+)$", out, re.M).groups()
        oc = re.search(rf"^own round {r} cutoff: (\S+) (\S+) (\S+)$", out, re.M).groups()
        assert rc[0] == oc[0]
        assert abs(float(rc[1]) - float(oc[1])) < 1e-5 * max(1.0, abs(float(rc[1])))


# ---------------------------------------------------------------------------
# binary checkpoint module (src/binary): dump a partition with CLVs, load it, evaluate
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["tv", "clv", "tv-repeats", "clv-repeats"])
def test_reference_binary_module_round_trip(oracle, tmp_path, mode):
    """the reference's checkpoint code walks the partition's arrays
    directly (src/binary/binary_io_operations.c:161-314); compiled unchanged against
    include/pll.h it must dump and restore a partition of this library such that the
    edge log-likelihood is the same without recomputing anything
    (test/src/binary/binary-sequential.c:344-354 uses 1e-7)."""
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this box")
    exe = tmp_path / "binary_driver"
    src = [os.path.join(ROOT, "tests", "dropin", "binary_driver.c"), f"{REF}/src/pllmod_common.c", f"{REF}/src/msa/pll_msa.c"]
    src += sorted(glob.glob(f"{REF}/src/binary/*.c"))
    inc = sum((["-I", d] for d in [f"{ROOT}/include", f"{REF}/src", f"{REF}/src/binary", f"{REF}/src/tree",
                                   f"{REF}/src/msa"]), [])
    libdir = os.path.dirname(ORACLE_LIB)
    # (pll_msa.c's statistics function calls into src/util, which is not on this path: unused sections are dropped)
    subprocess.run(["gcc", "-std=gnu99", "-D_GNU_SOURCE", "-O2", "-w", "-ffunction-sections", *inc, "-o", str(exe), *src,
                    "-L", libdir, "-lpll_oracle", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,--gc-sections"], check=True)
    out = subprocess.run([str(exe), str(tmp_path / "ckpt.bin"), mode], check=True, capture_output=True,
                         text=True, timeout=120).stdout
    v = {k.strip(): float(x) for k, x in re.findall(r"^(lnL \w+):\s+(-?[0-9.]+)$", out, re.M)}
    assert set(v) == {"lnL before", "lnL after", "lnL redo"}
    assert v["lnL before"] < -1000
    assert abs(v["lnL after"] - v["lnL before"]) < 1e-7
    assert abs(v["lnL redo"] - v["lnL before"]) < 1e-7
    assert "attributes restored: yes" in out
    # "-repeats": with PLL_ATTRIB_SITE_REPEATS the reference's dump / load and its empirical frequencies go through
    # partition->repeats (src/binary/binary_io_operations.c:231-282, src/msa/pll_msa.c:108-112): same numbers as without
    freqs = [float(x) for x in re.search(r"^freqs:(.*)$", out, re.M).group(1).split()]
    assert len(freqs) == 20 and abs(sum(freqs) - 1.0) < 1e-9
    seen = test_reference_binary_module_round_trip.__dict__.setdefault("seen", {})
    seen[mode] = (v["lnL before"], freqs)
    base = mode.split("-")[0]
    if base in seen and mode in seen and base != mode:
        assert seen[base] == seen[mode]
