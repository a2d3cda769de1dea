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
# round 3: every alphabet up to 64 states, mixtures, balanced trees, site repeats, partitions sharing launches
for trial in range(36):
    S = [17, 24, 32, 33, 48, 62, 64, 20, 4, 61, 28, 5][trial % 12]; R = 4
    n = int(rng.integers(5, 40)) if S <= 32 else int(rng.integers(5, 12)); N = int(rng.integers(1, 500))
    tr = pc.Tree(n, 300 + trial, 400 + trial, ladder=(trial % 3 == 1), balanced=(trial % 3 == 2))
    mix = [None, [0, 1, 0, 1], [0, 1, 2, 3]][(trial // 3) % 3]
    attrs = pc.PLL_ATTRIB_SITE_REPEATS if (S in (20, 4) and trial % 2 == 0) else 0
    kw = dict(states=S, rate_cats=R, ntips=n, nsites=N, coded=bool(trial % 5 != 4) or bool(attrs), tree=tr, mixture=mix,
              attributes=attrs)
    with pc.build_instance(product, **kw) as a, pc.build_instance(oracle, **{**kw, "attributes": 0}) as b:
        for rep in range(2):
            la, lb = pc.full_traversal(a, one_by_one_pmatrices=bool(rep)), pc.full_traversal(b, one_by_one_pmatrices=bool(rep))
            tol = (2e-6 if S > 32 else 1e-11) * abs(lb) + 2e-9 * N
            ok = abs(la - lb) <= tol and all(np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])) for op in tr.ops)
            if not ok: bad += 1; print("r3 trav", S, n, N, mix, attrs, la, lb)
for trial in range(6):
    n = int(rng.integers(6, 50))
    tr = pc.Tree(n, 500 + trial, 600 + trial, balanced=bool(trial % 2))
    spec = [(S, int(rng.integers(1, 3000))) for S in ([20, 20, 20, 4, 4, 10, 10, 61][:int(rng.integers(2, 9))])]
    def members(lib):
        out = []
        for k, (S, N) in enumerate(spec):
            m = pc.build_instance(lib, states=S, rate_cats=4, ntips=n, nsites=N, coded=True, tree=tr, seed_shift=0)
            m.tree = tr
            out.append(m)
        return out
    ga, gb, oc = members(product), members(product), members(oracle)
    for m in ga + gb + oc:
        m.update_pmatrices(np.arange(tr.nedges), tr.brlens)
    pc.update_partials_batch(product, ga, tr.ops_with_scalers(True))
    for m in gb + oc:
        m.update_partials(tr.ops_with_scalers(True))
    for x, y, z in zip(ga, gb, oc):
        sa, sb = tr.scaler_of(tr.root_a), tr.scaler_of(tr.root_b)
        lx, ly, lz = (i.edge_lnl(tr.root_a, sa, tr.root_b, sb, tr.root_matrix) for i in (x, y, z))
        if lx != ly or abs(lx - lz) > (2e-6 if x.S > 32 else 1e-11) * abs(lz) + 2e-9 * x.N:
            bad += 1; print("batch", spec, lx, ly, lz)
    for m in ga + gb + oc:
        m.close()
# site repeats over subtrees: simulated alignments (real repeats), random shapes and sizes, a second evaluation
# from another root edge (class nodes of the first one under new parents); on / off must agree exactly
for trial in range(40):
    S = [4, 20][trial % 2]
    n = int(rng.integers(4, 70)); N = int(rng.integers(50, 30_000))
    shape = dict(ladder=(trial % 5 == 1), balanced=(trial % 5 == 2), brlen_range=(0.005, float(rng.uniform(0.02, 0.3))))
    tr = pc.Tree(n, 700 + trial, 800 + trial, **shape)
    codes = pc.simulated_codes(tr, N, S, seed=900 + trial, scale=float(rng.uniform(0.3, 2.0)))
    edge = int(rng.integers(0, tr.nedges))
    res = []
    for attrs in (pc.PLL_ATTRIB_SITE_REPEATS, 0):
        with pc.build_instance(product, states=S, rate_cats=[4, 4, 2, 1][trial % 4], ntips=n, nsites=N, coded=True, tree=tr,
                               attributes=attrs, codes=codes) as a:
            a.tree = tr
            l1 = pc.full_traversal(a)
            sc = [a.get_scaler(op[1]).tobytes() for op in tr.ops]
            t2 = pc.Tree(n, 700 + trial, 800 + trial, **shape)
            t2.set_root_edge(edge)
            a.tree = t2
            l2 = pc.full_traversal(a)
            res.append((l1, l2, sc, [a.get_clv(op[0]).tobytes() for op in t2.ops[::4]], [a.get_scaler(op[1]).tobytes() for op in t2.ops]))
    if res[0] != res[1]:
        bad += 1; print("classes", S, n, N, res[0][:2], res[1][:2])
print("stress done, failures:", bad)
