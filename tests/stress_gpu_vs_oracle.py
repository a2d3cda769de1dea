#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep beyond the pytest suite: incremental evaluation walks with
random seeds / tree sizes and full traversals over random (states, rate count, taxa, sites,
tree shape, tip storage) combinations; prints the number of mismatches."""
import sys, os
sys.path.insert(0, "pll-modules_amd"); sys.path.insert(0, "tests")
import numpy as np
import pllhip_ctypes as pc
import test_eval_driver as t
product = pc.PllLib(pc.PRODUCT_LIB)
oracle = pc.PllLib(os.path.join("oracle", "_build", "libpll_oracle.so"))
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10, (int(sys.argv[1]) if len(sys.argv) > 1 else 10) + 20):
    g = t._random_walk(product, 40, seed, ntips=12 + seed % 30)
    c = t._random_walk(oracle, 40, seed, ntips=12 + seed % 30)
    err = max(abs(x - y) / abs(y) for x, y in zip(g, c))
    if err > 1e-9: bad += 1; print("walk seed", seed, err)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
for trial in range(42):
    S = [4, 20, 61, 7, 16, 2, 10][trial % 7]; R = [1, 2, 4, 4][(trial // 7) % 4]
    n = int(rng.integers(5, 60)) if S < 61 else int(rng.integers(5, 14)); N = int(rng.integers(1, 700))
    tr = pc.Tree(n, 100 + trial, 200 + trial, ladder=bool(trial % 2))
    kw = dict(states=S, rate_cats=R, ntips=n, nsites=N, coded=bool(trial % 4 != 3), tree=tr,
              attributes=pc.PLL_ATTRIB_RATE_SCALERS if trial % 5 == 4 else 0)
    with pc.build_instance(product, **kw) as a, pc.build_instance(oracle, **kw) as b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        tol = (2e-6 if S > 20 else 1e-11) * abs(lb) + 2e-9 * N
        ok = abs(la - lb) <= tol and all(np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])) for op in tr.ops)
        if not ok: bad += 1; print("trav", S, R, n, N, la, lb)
# sizes at which the schedule changes form (rounds of chains as grid rows / the whole traversal in one launch)
for S, N, n in ((4, 70_000, 14), (4, 900_000, 8), (20, 40_000, 12), (20, 120_000, 10), (16, 110_000, 9), (2, 150_000, 16), (10, 30_000, 11)):
    tr = pc.Tree(n, int(rng.integers(1, 1000)), int(rng.integers(1, 1000)))
    kw = dict(states=S, rate_cats=4, ntips=n, nsites=N, coded=True, tree=tr)
    with pc.build_instance(product, **kw) as a, pc.build_instance(oracle, **kw) as b:
        for rep in range(2):                                  # second pass: cached schedule
            la, lb = pc.full_traversal(a), pc.full_traversal(b)
            ok = abs(la - lb) <= 1e-11 * abs(lb) + 2e-9 * N and \
                all(np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])) for op in tr.ops)
            if not ok: bad += 1; print("big", S, N, n, la, lb)
print("stress done, failures:", bad)
