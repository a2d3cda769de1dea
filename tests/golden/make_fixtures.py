#!/usr/bin/env python3
"""Generate tests/golden/blopt_fixtures.json from the reference's own test
programs and golden outputs (run in the build container, where /root/reference
exists; the JSON it writes is what travels to the GPU box).

Only DATA is extracted: the numeric literals that the two self-contained
reference tests feed into the likelihood path
(test/src/optimize/blopt-minimal.c:36-90, blopt-5states.c:28-80) and the numbers
they are expected to print (test/out/optimize/blopt-*.out).  No reference code
is copied.
"""
import json
import os
import re

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))


def c_array(src, name):
    m = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", src, re.S)
    return [float(x) if re.search(r"[.eE]", x) else int(x, 0)
            for x in re.findall(r"[-+]?(?:0x[0-9a-fA-F]+|\d+\.?\d*(?:[eE][-+]?\d+)?)", m.group(1))]


def parse_out(path, states):
    txt = open(path).read()
    blocks = re.split(r"P-matrix for branch length ([0-9.]+)\n", txt)
    pm = []
    for k in range(1, len(blocks), 2):
        rows = [[float(x) for x in ln.split()] for ln in blocks[k + 1].split("\n")
                if re.match(r"^[+-]\d", ln.strip())]
        rows = rows[:4 * states]
        pm.append({"t": float(blocks[k]),
                   "P": [rows[r * states:(r + 1) * states] for r in range(4)]})
    out = {
        "pmatrices": pm,
        "lnl_initial": float(re.search(r"Initial Log-L: ([-0-9.]+)", txt).group(1)),
        "lnl_after_blo": float(re.search(r"recomputed after BL-opt: ([-0-9.]+)", txt).group(1)),
        "neg_lnl_returned": float(re.search(r"returned by BL-opt:\s+([-0-9.]+)", txt).group(1)),
        "brlens_after_blo": [float(x) for x in re.findall(
            r":([0-9.]+)", re.search(r"Tree \(optimized\): (.*)", txt).group(1))],
    }
    return out


def main():
    src = open(os.path.join(REF, "src/optimize/blopt-minimal.c")).read()
    minimal = {
        "states": 4, "sites": 4, "rate_cats": 4, "alpha": 0.841,
        "branch_lengths": c_array(src, "branch_lengths"),
        "frequencies": c_array(src, "frequencies"),
        "subst_params": c_array(src, "subst_params"),
        "tip_clv": [c_array(src, "tip1"), c_array(src, "tip2"), c_array(src, "tip3")],
        "blo": {"min": 1e-4, "max": 1e3, "tolerance": 1e-2, "smoothings": 1, "radius": 1},
        "expected": parse_out(os.path.join(REF, "out/optimize/blopt-minimal.out"), 4),
    }
    src = open(os.path.join(REF, "src/optimize/blopt-5states.c")).read()
    five = {
        "states": 5, "sites": 4, "rate_cats": 4, "alpha": 0.841,
        "branch_lengths": c_array(src, "branch_lengths"),
        "frequencies": c_array(src, "frequencies"),
        "subst_params": c_array(src, "subst_params"),
        "charmap": c_array(src, "odd_map"),
        "sequences": re.findall(r'pll_set_tip_states \(partition, \d, odd_map, "([A-Z]+)"\)', src),
        "blo": {"min": 1e-4, "max": 1e3, "tolerance": 1e-4, "smoothings": 1, "radius": 1},
        "expected": parse_out(os.path.join(REF, "out/optimize/blopt-5states.out"), 5),
    }
    assert len(minimal["tip_clv"][0]) == 64 and len(five["charmap"]) == 256
    assert len(minimal["expected"]["pmatrices"]) == 3 and len(five["sequences"]) == 3
    with open(os.path.join(HERE, "blopt_fixtures.json"), "w") as f:
        json.dump({"blopt-minimal": minimal, "blopt-5states": five}, f, indent=1)


if __name__ == "__main__":
    main()
