"""HIP engine vs CPU oracle through the C ABI, on the same seeded inputs.

Tolerances: scalers, tip codes and invariant-site indices are integers and must
be bit-exact.  Floating point follows BASELINE.json's north star
(|dlnL| < 1e-6 per site); the tests hold the engine to far tighter bounds.  Up to 20 states:
lnL 1e-12 relative or 2e-9 per site, CLV / sumtable entries 1e-8 of their vector's maximum
(errors accumulate along the tree depth), derivatives 1e-9.  61 states: every engine runs its
OWN eigen-solver (nothing is injected); codon P-matrices hold entries down to 1e-19 which fp64
storage resolves to 1e-16 ABSOLUTE at best, and with random sequences whole sites consist of
such entries -- tests/test_expm_fixtures.py measures both engines against 60-digit matrix
exponentials (oracle 1.3e-16, product 2.7e-15 in P).  Hence for 61 states: lnL 5e-8 per site,
CLV entries 1e-7 of the largest entry of their SITE (all rates: the quantity the site
likelihood and the scaling rule see), derivatives 2e-6.  Measured deviations go to
gpurun_out/parity_measured.jsonl (condensed into profiles/r02_parity.json).
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import pllhip_ctypes as pc

pytestmark = pytest.mark.gpu
NONE = pc.PLL_SCALE_BUFFER_NONE
REL_CLV = 1e-8
REL_LNL = 1e-12
PER_SITE = 2e-9


PER_SITE_61 = 5e-8     # measured <= 7e-9 (profiles/r02_parity.json)
CLV_SITE_61 = 1e-7     # measured <= 5e-9


def lnl_close(la, lb, nsites, states=4):
    if states > 20:
        return abs(la - lb) <= PER_SITE_61 * nsites
    return abs(la - lb) <= max(REL_LNL * abs(lb), PER_SITE * nsites)


def site_err(a, b):
    """largest deviation of a CLV [site][rate][state], measured against the largest entry of the
    same SITE over all rates and states"""
    scale = np.maximum(np.abs(b).max(axis=(1, 2), keepdims=True), 1e-300)
    return float(np.max(np.abs(a - b) / scale))


def record(test, **values):
    """measured deviations -> gpurun_out/parity_measured.jsonl"""
    import json
    d = os.path.join(common.ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": test, **{k: float(v) for k, v in values.items()}}) + "\n")


def _pair(product, oracle, **kw):
    a = pc.build_instance(product, **kw)
    b = pc.build_instance(oracle, **kw, tree=a.tree)
    return a, b


def _compare_full(a, b, check_clvs=True):
    la = pc.full_traversal(a)
    lb = pc.full_traversal(b)
    assert np.isfinite(lb) and lb < 0
    assert lnl_close(la, lb, a.N, a.S), (la, lb, abs(la - lb) / a.N)
    worst = 0.0
    if check_clvs:
        t = a.tree
        for op in t.ops:
            ca, cb = a.get_clv(op[0]), b.get_clv(op[0])
            if a.S <= 20:
                err = common.vec_err(ca, cb)
                assert err < REL_CLV, f"CLV {op[0]}"
            else:
                err = site_err(ca, cb)
                assert err < CLV_SITE_61, f"CLV {op[0]}: {err}"
            worst = max(worst, err)
            if a.nscalers:
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])), f"scaler {op[1]}"
    if a.S > 20:
        record("full_traversal_61", sites=a.N, rate_cats=a.R, dlnl_per_site=abs(la - lb) / a.N, clv_site_err=worst)
    return la, lb


def test_device_is_gfx950(product):
    buf = C.create_string_buffer(64)
    assert product.lib.pllhip_device_arch(0, buf, 64)
    assert buf.value.decode().startswith("gfx950")


@pytest.mark.parametrize("name,coded", [("blopt-minimal", False), ("blopt-5states", False),
                                        ("blopt-5states", True)])
def test_golden_fixtures_on_gpu(product, name, coded):
    """P-matrices, lnL and the Newton-Raphson post-optimisation numbers of the
    reference's golden files, computed by the HIP kernels"""
    expected, got = common.run_golden_case(product, name, coded=coded)
    common.check_golden(expected, got)


@pytest.mark.parametrize("states,rate_cats", [(4, 4), (4, 1), (4, 2), (4, 8), (4, 16), (4, 3),
                                              (20, 4), (20, 1), (20, 2), (20, 8), (20, 12),
                                              (5, 4), (2, 3), (7, 4), (61, 2), (61, 4), (61, 1),
                                              (2, 4), (3, 4), (10, 4), (16, 4), (16, 1), (13, 5), (8, 2), (4, 5),
                                              (17, 4), (33, 2)])
@pytest.mark.parametrize("coded", [True, False])
def test_full_traversal_parity(product, oracle, states, rate_cats, coded):
    ntips, nsites = (9, 257) if states > 20 else (14, 1031)
    a, b = _pair(product, oracle, states=states, rate_cats=rate_cats, ntips=ntips, nsites=nsites,
                 coded=coded)
    with a, b:
        _compare_full(a, b)


@pytest.mark.parametrize("states", [4, 20, 5, 61, 10, 2])
@pytest.mark.parametrize("nsites", [1, 2, 63, 64, 65, 255, 256, 1000, 4097])
def test_ragged_site_counts(product, oracle, states, nsites):
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=6, nsites=nsites, coded=True)
    with a, b:
        _compare_full(a, b)


@pytest.mark.parametrize("states,ncodes", [(20, 45), (4, 15), (61, 70), (7, 40)])
def test_many_ambiguity_codes(product, oracle, states, ncodes):
    """tip alphabets with many ambiguity codes: beyond 32 (20 states) / 67 (61 states)
    codes the lookup tables no longer fit the LDS staging area and are gathered from
    global memory instead; code tables also grow after P-matrices exist (LUT rebuild)"""
    ntips, nsites = 9, 333
    rng = pc.splitmix64(1234 + states, ncodes * 3)
    cmap = np.zeros(256, dtype=np.uint64)
    full = (1 << states) - 1
    for c in range(ncodes):
        if c < states:
            m = 1 << c
        else:
            m = (1 << int(rng[3 * c] % np.uint64(states))) | (1 << int(rng[3 * c + 1] % np.uint64(states))) | \
                (1 << int(rng[3 * c + 2] % np.uint64(states)))
        cmap[48 + c] = m & full
    masks = {int(cmap[48 + c]) for c in range(ncodes)}
    seqs = (pc.splitmix64(77, ntips * nsites) % np.uint64(ncodes)).astype(np.uint8).reshape(ntips, nsites)
    insts = []
    for lib in (product, oracle):
        inst = pc.build_instance(lib, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=True)
        l_plain = pc.full_traversal(inst)           # P-matrices + LUTs exist for the plain alphabet
        for t in range(ntips):
            inst.set_tip_states(t, cmap, (seqs[t] + 48).tobytes())
        insts.append((inst, l_plain))
    (a, la0), (b, lb0) = insts
    with a, b:
        assert a.p.contents.maxstates == b.p.contents.maxstates >= min(len(masks), ncodes) - 1
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert lnl_close(la, lb, nsites, states), (la, lb)
        assert abs(la - la0) > 1.0                   # the ambiguous alignment really is different
        t = a.tree
        sa, sb = a.alloc_sumtable(), b.alloc_sumtable()
        args = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        a.update_sumtable(*args, sa); b.update_sumtable(*args, sb)
        assert np.allclose(a.derivatives(args[2], args[3], 0.2, sa), b.derivatives(args[2], args[3], 0.2, sb),
                           rtol=1e-8 if states <= 20 else 2e-6)
        a.free_sumtable(sa); b.free_sumtable(sb)


def test_empty_partition(product):
    with pc.Instance(product, 3, 4, 0, 4, scalers=True) as a:
        a.set_model(pc.DNA_GTR_RATES, pc.DNA_FREQS, product.gamma_cats(1.0, 4))
        a.update_pmatrices([0, 1, 2], [0.1, 0.2, 0.3])
        a.update_partials([(3, 0, 0, 0, NONE, 1, 1, NONE)])
        assert a.edge_lnl(3, 0, 2, NONE, 2) == 0.0


@pytest.mark.parametrize("states,ntips,rate_cats", [(4, 600, 4), (20, 260, 4), (61, 130, 4), (61, 130, 1),
                                                    (10, 350, 4), (2, 900, 3), (16, 250, 2)])
def test_deep_tree_scaling_is_bit_exact(product, oracle, states, ntips, rate_cats):
    """random sequences on a deep tree drive CLVs below 2^-256: scaler counts must
    agree exactly and lnL must survive"""
    a, b = _pair(product, oracle, states=states, rate_cats=rate_cats, ntips=ntips, nsites=97, coded=True)
    with a, b:
        # 61 states: the improbable components of a vector drift apart with the depth (each
        # engine's own 1e-15 in the small P-matrix entries, 130 levels): this test checks the
        # scaler counts (exact) and lnL, not the small components
        la, lb = _compare_full(a, b, check_clvs=states <= 20)
        if states > 20:
            for op in a.tree.ops:
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
        root_sc = a.get_scaler(a.tree.scaler_of(a.tree.root_a))
        assert root_sc.max() >= 1, "test did not reach the scaling regime"


def test_codon_launch_modes_agree_bitwise(product):
    """61 states: small slices are launched one workgroup per (range, rate) with the scaling
    votes combined -- and predicted from the previous evaluation -- by a second kernel;
    large ones walk the rates inside one workgroup.  Same numbers either way, also on a
    second evaluation (when the predictions are in use) and after a model change."""
    import subprocess
    import sys
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import pllhip_ctypes as pc
lib = pc.PllLib(pc.PRODUCT_LIB)
with pc.build_instance(lib, states=61, rate_cats=4, ntips=130, nsites=97, coded=True) as a:
    out = []
    for rep in range(3):
        if rep == 2:
            subst, freqs = pc.codon_model()
            a.set_model(subst, freqs, lib.gamma_cats(2.5, 4))
            a.update_pmatrices(np.arange(a.tree.nedges), a.tree.brlens)
        l = pc.full_traversal(a)
        h = hashlib.sha256()
        for op in a.tree.ops:
            h.update(a.get_clv(op[0]).tobytes())
            h.update(a.get_scaler(op[1]).tobytes())
        out.append("%%.17g %%s" %% (l, h.hexdigest()))
    assert a.get_scaler(a.tree.scaler_of(a.tree.root_a)).max() >= 1
    print("\n".join(out))
""" % os.path.dirname(pc.__file__)
    runs = []
    # rate-parallel with cherries folded into their consumers, rate-parallel without, rates in a workgroup
    for mode, cherries in (("1", "1"), ("1", "0"), ("0", "1")):
        env = dict(os.environ, PLLHIP_S61_RATEPAR=mode, PLLHIP_S61_CHERRIES=cherries)
        runs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True,
                                   text=True, timeout=300).stdout)
    assert runs[0] == runs[1] == runs[2] and runs[0].count("\n") == 3
    first, second, third = runs[0].splitlines()
    assert first == second and third != first


@pytest.mark.parametrize("states,ntips", [(4, 300), (20, 120), (10, 150), (2, 400)])
def test_traversal_schedules_agree_bitwise(product, states, ntips):
    """4 / 20 / 2..16 states: a partition that has its device to itself runs a whole traversal as ONE
    launch (waves keep their sites through every chain); otherwise one launch per round of chains (per
    dependency level at 2..16 states).  Same
    vectors, scalers and lnL either way -- full traversals, a second evaluation (cached schedule), a
    partial traversal and a re-rooted one."""
    import subprocess
    import sys
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import pllhip_ctypes as pc
lib = pc.PllLib(pc.PRODUCT_LIB)
import os
shape = os.environ.get("TEST_TREE_SHAPE", "random")
tree = pc.Tree(%d, 42, 43, ladder=(shape == "ladder"), balanced=(shape == "balanced"))
with pc.build_instance(lib, states=%d, rate_cats=4, ntips=tree.ntips, nsites=1531, coded=True, tree=tree) as a:
    out = []
    t = a.tree
    for rep in range(4):
        if rep == 2:
            t.set_root_edge(3)
        if rep == 3:                          # partial traversal: the last third of the operations
            ops = t.ops_with_scalers(True)
            part = ops[2 * len(ops) // 3:]
            a.update_partials(a.make_ops(part), len(part))
            l = a.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
        else:
            l = pc.full_traversal(a)
        h = hashlib.sha256()
        for op in t.ops:
            h.update(a.get_clv(op[0]).tobytes())
            h.update(a.get_scaler(op[1]).tobytes())
        out.append("%%.17g %%s" %% (l, h.hexdigest()))
    print("\n".join(out))
    print(a.counters().partial_launches)
""" % (os.path.dirname(pc.__file__), ntips, states)
    for shape in ("random", "balanced"):
        runs = []
        for mode in ("1", "0"):
            env = dict(os.environ, PLLHIP_TRAVERSE=mode, TEST_TREE_SHAPE=shape)
            runs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True,
                                       text=True, timeout=300).stdout.splitlines())
        assert runs[0][:4] == runs[1][:4] and len(runs[0]) == 5
        if not common.FORCED_TRANSIENT:       # (vectors that were not stored are recomputed for the hashes: more launches)
            assert int(runs[0][4]) == 4 and int(runs[1][4]) > 8        # one launch per traversal / several


@pytest.mark.parametrize("shape", ["random", "ladder", "balanced"])
@pytest.mark.parametrize("states", [4, 20, 10, 61])
def test_tree_shapes_against_the_oracle(product, oracle, states, shape):
    """the chain / round / one-launch schedules are cut from the tree: a caterpillar (one chain after the other),
    a complete binary tree (half of the operations are cherries, 13 levels for 100 taxa) and the random
    stepwise-addition trees of the benchmarks; vectors, scaler counts and lnL against the oracle, operation by
    operation, with site repeats on top at 20 states"""
    ntips, nsites = (40, 130) if states > 20 else (100, 777)
    tree = pc.Tree(ntips, 42, 43, ladder=(shape == "ladder"), balanced=(shape == "balanced"))
    for attributes in ((0, pc.PLL_ATTRIB_SITE_REPEATS) if states == 20 else (0,)):
        a = pc.build_instance(product, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=True, tree=tree,
                              attributes=attributes)
        b = pc.build_instance(oracle, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=True, tree=tree)
        with a, b:
            # (61 states: a cherry site whose two codons differ in all three positions is made of P-matrix entries
            # at the 1e-16 absolute floor of the two eigen-solvers -- 2e-7 of that site's largest entry, the class of
            # deviation tests/test_expm_fixtures.py pins; vectors are compared up to 20 states, scaler counts and lnL always)
            _compare_full(a, b, check_clvs=states <= 20)
            for op in tree.ops:
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
            _compare_full(a, b, check_clvs=False)          # the resident schedule once more


@pytest.mark.parametrize("states,nsites,ntips,launches", [(4, 900_000, 8, 1), (4, 70_000, 14, None), (20, 120_000, 10, 1),
                                                          (20, 40_000, 12, None), (16, 110_000, 9, 1), (2, 150_000, 16, None)])
def test_schedules_at_the_sizes_that_select_them(product, oracle, states, nsites, ntips, launches):
    """the form of a traversal follows the size of the partition (rounds of chains as grid rows / everything
    in one launch): both forms at sizes that select them by default, against the oracle (lnL, exact scaler
    counts), twice (the second pass replays the resident schedule)"""
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=True)
    with a, b:
        for rep in range(2):
            before = a.counters().partial_launches
            la, lb = pc.full_traversal(a), pc.full_traversal(b)
            used = a.counters().partial_launches - before
            assert abs(la - lb) <= 1e-11 * abs(lb) + 2e-9 * nsites, (la, lb)
            for op in a.tree.ops:
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
            if common.FORCED:
                continue                      # (class operations / recomputations change the launch counts)
            if launches is not None:
                assert used == launches
            else:
                assert 1 < used < len(a.tree.ops)


@pytest.mark.parametrize("rate_cats,attributes", [(4, 0), (2, 0), (4, pc.PLL_ATTRIB_RATE_SCALERS)])
def test_codon_cherries_that_scale(product, oracle, rate_cats, attributes):
    """61 states: a tip x tip operation is computed inside the operation that consumes it, its own
    scaling decided per pair of tip codes.  A cherry only ever scales when its entries are exact zeros
    (P-matrix noise is 1e-17, far above 2^-256): pendant branches of length 0 -- identity matrices --
    make every site with two different codons an all-zero, scaled site.  Scalers (exact) and vectors
    must be what the oracle computes operation by operation; lnL is -inf on both sides."""
    a, b = _pair(product, oracle, states=61, rate_cats=rate_cats, ntips=24, nsites=203, coded=True,
                 attributes=attributes)
    with a, b:
        t = a.tree
        cherries = [op for op in t.ops if op[2] < t.ntips and op[5] < t.ntips]
        assert len(cherries) >= 3
        for k, op in enumerate(cherries):        # the pair shares the tree: both engines see these lengths
            if k % 3 != 2:                       # two of three cherries lose their pendant branches
                t.brlens[op[3]] = 0.0
                t.brlens[op[6]] = 0.0 if k % 3 == 0 else 0.05
        for rep in range(2):                     # second pass: scaling predictions of the consumers in use
            la, lb = pc.full_traversal(a), pc.full_traversal(b)
            assert la == lb == -np.inf
            for op in t.ops:
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])), f"scaler {op[1]}"
                # (with per-rate scalers a rate that carries only P-matrix noise -- 1e-18 through a zero-length
                # branch -- is scaled up until it is the largest entry of its site: the vectors above the
                # cherries then differ by the two engines' noise, with or without the folding; the counts
                # are exact)
                if not attributes or op in cherries:
                    assert site_err(a.get_clv(op[0]), b.get_clv(op[0])) < CLV_SITE_61, f"CLV {op[0]}"
        scaled = sum(int(b.get_scaler(op[1]).sum()) for op in cherries)
        assert scaled > 0, "no cherry site reached the scaling regime"


def test_scaling_on_equals_scaling_off(product):
    kw = dict(states=4, rate_cats=4, ntips=30, nsites=301, coded=True)
    a = pc.build_instance(product, scalers=True, **kw)
    b = pc.build_instance(product, scalers=False, **kw)
    with a, b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la - lb) < 1e-9 * abs(la)


@pytest.mark.parametrize("states", [4, 20, 5])
def test_rerooting_invariance(product, states):
    with pc.build_instance(product, states=states, rate_cats=4, ntips=11, nsites=513, coded=True) as a:
        ref = pc.full_traversal(a)
        for k in (0, 3, 7, a.tree.nedges - 1):
            a.tree.set_root_edge(k)
            assert abs(pc.full_traversal(a) - ref) < 1e-10 * abs(ref)


@pytest.mark.parametrize("states", [4, 20, 5, 61])
def test_persite_and_pattern_weights(product, oracle, states):
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=8, nsites=333, coded=True)
    with a, b:
        w = (pc.splitmix64(5, 333) % np.uint64(7)).astype(np.uint32)
        a.set_pattern_weights(w)
        b.set_pattern_weights(w)
        pc.full_traversal(a); pc.full_traversal(b)
        t = a.tree
        la, pa = a.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b),
                            t.root_matrix, persite=True)
        lb, pb = b.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b),
                            t.root_matrix, persite=True)
        assert np.allclose(pa, pb, rtol=1e-12 if states <= 20 else 1e-9, atol=0)
        assert abs(la - float(np.dot(pa, w))) < 1e-9 * abs(la)
        assert lnl_close(la, lb, a.N)


@pytest.mark.parametrize("states", [4, 20, 5, 61, 2, 10, 16])
def test_root_loglikelihood(product, oracle, states):
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=8, nsites=129, coded=False)
    with a, b:
        pc.full_traversal(a); pc.full_traversal(b)
        t = a.tree
        ra = a.root_lnl(t.root_a, t.scaler_of(t.root_a))
        rb = b.root_lnl(t.root_a, t.scaler_of(t.root_a))
        assert abs(ra - rb) < REL_LNL * abs(rb)


@pytest.mark.parametrize("states,coded", [(4, True), (4, False), (20, True), (5, True), (20, False),
                                          (61, True), (61, False), (2, True), (10, True), (10, False), (16, False)])
def test_sumtable_and_derivatives(product, oracle, states, coded):
    nsites = 515 if states <= 20 else 131
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=10, nsites=nsites, coded=coded)
    with a, b:
        pc.full_traversal(a); pc.full_traversal(b)
        t = a.tree
        # eigenvectors are unique only up to sign and order, so sumtable entries of two solvers
        # are not comparable one by one; what is: sum_k sumtable[n,r,k] exp(lambda_k x) -- the
        # likelihood of (site, rate) as a function of the branch length -- with each engine's
        # own eigenvalues
        la_ = np.ctypeslib.as_array(a.p.contents.eigenvals[0], shape=(a.Sp,))[:a.S].copy()
        lb_ = np.ctypeslib.as_array(b.p.contents.eigenvals[0], shape=(b.Sp,))[:b.S].copy()
        sa_, sb_ = a.alloc_sumtable(), b.alloc_sumtable()
        for (pc_, cc_) in ((t.root_a, t.root_b), (t.root_b, t.root_a)):
            if coded and pc_ < t.ntips and cc_ < t.ntips:
                continue
            args = (pc_, cc_, t.scaler_of(pc_), t.scaler_of(cc_))
            a.update_sumtable(*args, sa_)
            b.update_sumtable(*args, sb_)
            ta, tb = a.get_sumtable(sa_), b.get_sumtable(sb_)
            for x in (0.0, 0.05, 0.7):
                fa, fb = ta @ np.exp(la_ * x), tb @ np.exp(lb_ * x)          # [site][rate]
                scale = np.abs(fb).max(axis=1, keepdims=True)
                # 61 states: products of two CLVs that each carry up to 2e-6 of their site's maximum
                assert (np.abs(fa - fb) / scale).max() < (1e-9 if states <= 20 else 2e-5)
            for bl in (1e-4, 0.013, 0.1, 0.77, 5.0, 90.0):
                da = a.derivatives(args[2], args[3], bl, sa_)
                db = b.derivatives(args[2], args[3], bl, sb_)
                assert np.allclose(da, db, rtol=1e-9 if states <= 20 else 2e-6, atol=1e-9 * a.N), (bl, da, db)
        a.free_sumtable(sa_); b.free_sumtable(sb_)


@pytest.mark.parametrize("states", [4, 20])
def test_derivatives_match_finite_differences(product, states):
    with pc.build_instance(product, states=states, rate_cats=4, ntips=7, nsites=200, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        st = a.alloc_sumtable()
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        a.update_sumtable(t.root_a, t.root_b, sa, sb, st)
        x, h = 0.2, 1e-5

        def neg_lnl(bl):
            a.update_pmatrices([t.root_matrix], [bl])
            return -a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
        df, ddf = a.derivatives(sa, sb, x, st)
        f0, fp, fm = neg_lnl(x), neg_lnl(x + h), neg_lnl(x - h)
        assert abs(df - (fp - fm) / (2 * h)) < 1e-5 * max(1.0, abs(df))
        assert abs(ddf - (fp - 2 * f0 + fm) / (h * h)) < 2e-3 * max(1.0, abs(ddf))
        a.free_sumtable(st)


@pytest.mark.parametrize("states", [4, 20, 5, 2, 10])
def test_invariant_sites(product, oracle, states):
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=5, nsites=4000, coded=True)
    with a, b:
        assert product.lib.pll_update_invariant_sites(a.p) and oracle.lib.pll_update_invariant_sites(b.p)
        ia = np.ctypeslib.as_array(a.p.contents.invariant, shape=(a.N,))
        ib = np.ctypeslib.as_array(b.p.contents.invariant, shape=(b.N,))
        assert np.array_equal(ia, ib)
        if states <= 5:
            assert (ib >= 0).any(), "no invariant column in the sample"
        a.set_pinv(0.25); b.set_pinv(0.25)
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert lnl_close(la, lb, a.N)
        t = a.tree
        sa_, sb_ = a.alloc_sumtable(), b.alloc_sumtable()
        args = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        a.update_sumtable(*args, sa_); b.update_sumtable(*args, sb_)
        da = a.derivatives(args[2], args[3], 0.1, sa_)
        db = b.derivatives(args[2], args[3], 0.1, sb_)
        assert np.allclose(da, db, rtol=1e-8)
        a.free_sumtable(sa_); b.free_sumtable(sb_)


@pytest.mark.parametrize("states", [4, 20, 5, 61, 10, 2])
def test_node_ancestral_states(product, oracle, states):
    """marginal ancestral state probabilities (src/tree/treeinfo.c:1698)"""
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=7, nsites=77, coded=True)
    with a, b:
        pc.full_traversal(a); pc.full_traversal(b)
        t = a.tree
        args = (t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
        pa, pb = a.node_ancestral(*args), b.node_ancestral(*args)
        assert np.allclose(pa.sum(axis=1), 1.0, atol=1e-12)
        assert np.allclose(pa, pb, rtol=1e-9, atol=1e-12 if states <= 20 else 2e-6)


def test_host_model_arrays_are_source_of_truth(product, oracle):
    """pll-modules writes partition->rates / frequencies / subst_params directly and
    flips eigen_decomp_valid (src/algorithm/algo_callback.c:44-68): the engine must
    notice on the next kernel-entry call"""
    a, b = _pair(product, oracle, states=4, rate_cats=4, ntips=8, nsites=300, coded=True)
    with a, b:
        l0 = pc.full_traversal(a)
        for inst in (a, b):
            p = inst.p.contents
            for r in range(4):
                p.rates[r] = [0.2, 0.6, 1.1, 2.1][r]
            p.frequencies[0][0], p.frequencies[0][3] = 0.3, 0.26
            p.subst_params[0][1] = 2.5
            p.eigen_decomp_valid[0] = 0
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la - l0) > 1e-3
        assert lnl_close(la, lb, a.N)
        assert a.counters().model_uploads >= 2


def test_one_by_one_pmatrix_calls_and_single_op_calls(product):
    """the reference's call granularity: one branch per pll_update_prob_matrices call
    (src/tree/treeinfo.c:845-865) and one op per pll_update_partials call
    (src/optimize/pll_optimize.c:773) must give the same numbers as batched calls"""
    with pc.build_instance(product, states=20, rate_cats=4, ntips=12, nsites=300, coded=True) as a:
        ref = pc.full_traversal(a)
        t = a.tree
        a.update_pmatrices(np.arange(t.nedges), t.brlens, one_by_one=True)
        for op in t.ops:
            a.update_partials([op])
        got = a.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
        assert got == ref
        c = a.counters()
        # (under PLLHIP_TRANSIENT=1 vectors the first traversal kept in registers are recomputed for the single operations)
        assert c.partial_ops == 2 * len(t.ops) or common.FORCED_TRANSIENT


@pytest.mark.parametrize("states,ladder,rate_cats", [(4, False, 4), (4, True, 4), (20, True, 4), (4, False, 1),
                                                     (4, True, 2), (20, False, 1), (20, True, 2)])
def test_chained_launches_store_what_single_operations_store(product, oracle, states, ladder, rate_cats):
    """a whole op list goes out as operation chains (the vector of a link stays in
    registers, engine.h ChainBatch); one-op calls cannot chain.  Both must leave
    bit-identical CLVs and scalers -- on a caterpillar tree too, whose single long path
    is cut into several chains -- and agree with the oracle in the scaling regime."""
    n = 400 if states == 4 else 90
    t = pc.Tree(n, 42, 43, brlen_range=(0.05, 0.6), ladder=ladder)
    kw = dict(states=states, rate_cats=rate_cats, ntips=n, nsites=197, coded=True, tree=t)
    with pc.build_instance(product, **kw) as a, pc.build_instance(product, **kw) as b, \
            pc.build_instance(oracle, **kw) as o:
        la = pc.full_traversal(a)
        launches = a.counters().partial_launches
        b.update_pmatrices(np.arange(t.nedges), t.brlens)
        for op in t.ops:
            b.update_partials([op])
        lb = b.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
        assert la == lb
        assert launches < len(t.ops) // 2                      # chains, not levels of single ops
        lo = pc.full_traversal(o)
        assert lnl_close(la, lo, a.N)
        for op in t.ops:
            assert np.array_equal(a.get_clv(op[0]), b.get_clv(op[0])), f"CLV {op[0]}"
            assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1])), f"scaler {op[1]}"
            assert np.array_equal(a.get_scaler(op[1]), o.get_scaler(op[1])), f"scaler {op[1]} vs oracle"
        assert a.get_scaler(t.scaler_of(t.root_a)).max() >= 1, "test did not reach the scaling regime"


def test_run_to_run_determinism(product):
    """SPR rounds assert reproducibility (src/algorithm/algo_search.c:1453-1457)"""
    with pc.build_instance(product, states=20, rate_cats=4, ntips=20, nsites=5000, coded=True) as a:
        vals = {pc.full_traversal(a) for _ in range(4)}
        assert len(vals) == 1


MATRIX_CORE_ALPHABETS = [17, 24, 25, 32, 33, 48, 60, 62, 63, 64]


@pytest.mark.parametrize("states", MATRIX_CORE_ALPHABETS)
@pytest.mark.parametrize("coded", [True, False])
def test_every_alphabet_up_to_64_states_runs_on_the_matrix_cores(product, oracle, states, coded):
    """multistate alphabets up to 64 states (src/util/models_mult.c:92-97) and the 60 / 62 / 63-codon genetic
    codes: no state count falls back to the one-thread-per-element kernels; CLVs, scalers, lnL, root lnL,
    sumtable and derivatives (single and multi-length) against the oracle, with and without per-rate scalers"""
    name = None
    for attributes in (0, pc.PLL_ATTRIB_RATE_SCALERS):
        ntips, nsites = (9, 257) if states > 32 else (14, 517)
        a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=coded,
                     attributes=attributes)
        with a, b:
            name = product.lib.pllhip_partials_kernel_name(a.p).decode()
            assert name != "generic", (states, name)
            la, lb = pc.full_traversal(a), pc.full_traversal(b)
            tol = 5e-8 * a.N if states > 32 else max(REL_LNL * abs(lb), PER_SITE * a.N)
            assert abs(la - lb) <= tol, (la, lb)
            t = a.tree
            for op in t.ops:
                ca, cb = a.get_clv(op[0]), b.get_clv(op[0])
                err = site_err(ca, cb)
                assert err < (1e-8 if states <= 32 else CLV_SITE_61), (op[0], err)
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
            sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
            ra, rb = a.root_lnl(t.root_a, sa), b.root_lnl(t.root_a, sa)
            assert abs(ra - rb) <= tol
            sta, stb = a.alloc_sumtable(), b.alloc_sumtable()
            a.update_sumtable(t.root_a, t.root_b, sa, sb, sta)
            b.update_sumtable(t.root_a, t.root_b, sa, sb, stb)
            lens = [1e-3, 0.05, 0.4, 3.0]
            rtol = 1e-9 if states <= 32 else 1e-5
            mdf, mddf = a.derivatives_multi(sa, sb, lens, sta)
            for k, bl in enumerate(lens):
                da, db = a.derivatives(sa, sb, bl, sta), b.derivatives(sa, sb, bl, stb)
                assert np.allclose(da, db, rtol=rtol, atol=1e-9 * a.N), (bl, da, db)
                assert np.allclose([mdf[k], mddf[k]], db, rtol=rtol, atol=1e-9 * a.N)
            a.free_sumtable(sta); b.free_sumtable(stb)


@pytest.mark.parametrize("states,ntips", [(24, 300), (32, 300), (48, 140), (64, 140)])
def test_deep_trees_scale_bit_exactly_at_the_new_alphabets(product, oracle, states, ntips):
    a, b = _pair(product, oracle, states=states, rate_cats=4, ntips=ntips, nsites=97, coded=True)
    with a, b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la - lb) <= (5e-8 if states > 32 else 2e-9) * a.N * max(1.0, ntips / 50)
        top = 0
        for op in a.tree.ops:
            sa_, sb_ = a.get_scaler(op[1]), b.get_scaler(op[1])
            assert np.array_equal(sa_, sb_)
            top = max(top, int(sa_.max()))
        assert top >= 1


def test_specialised_kernels_are_the_ones_running(product):
    for states, name in ((4, b"s4-valu"), (20, b"s20-mfma"), (61, b"s61-mfma"), (5, b"s16-mfma"), (2, b"s16-mfma"),
                         (10, b"s16-mfma"), (16, b"s16-mfma"), (17, b"s16-mfma"), (32, b"s16-mfma"), (33, b"s61-mfma"),
                         (62, b"s61-mfma"), (64, b"s61-mfma")):
        with pc.Instance(product, 3, states, 8, 4) as a:
            assert product.lib.pllhip_partials_kernel_name(a.p) == name
    with pc.Instance(product, 3, 4, 8, 3) as a:                      # 4 states, odd rate count
        assert product.lib.pllhip_partials_kernel_name(a.p) == b"s16-mfma"
    with pc.Instance(product, 3, 32, 8, 16) as a:                    # the tables of 16 rates do not fit the LDS
        assert product.lib.pllhip_partials_kernel_name(a.p) == b"generic"
    for states, name in ((4, b"s16-mfma"), (10, b"s16-mfma"), (20, b"s20-mfma"), (61, b"s61-mfma"), (33, b"s61-mfma")):
        with pc.Instance(product, 3, states, 8, 4, attributes=pc.PLL_ATTRIB_RATE_SCALERS) as a:
            assert product.lib.pllhip_partials_kernel_name(a.p) == name


def test_error_reporting(product):
    with pc.Instance(product, 3, 4, 8, 4, scalers=False, clv_buffers=1, prob_matrices=3) as a:
        a.set_model(pc.DNA_GTR_RATES, pc.DNA_FREQS, product.gamma_cats(1.0, 4))
        mi = np.array([7], dtype=np.uint32); bl = np.array([0.1])
        assert not product.lib.pll_update_prob_matrices(a.p, a.params_p, mi.ctypes.data_as(pc.c_uint_p),
                                                        bl.ctypes.data_as(pc.c_double_p), 1)
        assert product.errno == 113
        with pytest.raises(RuntimeError):
            a.update_partials([(9, NONE, 0, 0, NONE, 1, 1, NONE)])
        st = a.alloc_sumtable()
        with pytest.raises(RuntimeError):
            a.derivatives(NONE, NONE, 0.1, st)      # no sumtable computed for this key
        a.free_sumtable(st)


def test_tiled_alignment_scales_linearly(product):
    """size-independent property at a large site count: an alignment tiled K times
    has exactly K times the lnL of one tile (same per-site arithmetic, and the
    final sum is exact to rounding)"""
    base = pc.build_instance(product, states=4, rate_cats=4, ntips=16, nsites=1000, coded=True)
    K = 300
    with base:
        l1 = pc.full_traversal(base)
        big = pc.Instance(product, 16, 4, 1000 * K, 4, attributes=pc.PLL_ATTRIB_PATTERN_TIP)
        with big:
            big.set_model(pc.DNA_GTR_RATES, pc.DNA_FREQS, product.gamma_cats(0.841, 4))
            cmap = pc.state_charmap(4)
            for t in range(16):
                big.set_tip_states(t, cmap, (np.tile(base.codes[t], K) + 48).tobytes())
            big.tree = base.tree
            lk = pc.full_traversal(big)
            assert abs(lk - K * l1) < 1e-10 * abs(lk)


@pytest.mark.parametrize("cfg,tile", [("c2", 1000), ("c3", 1000), ("c5", 500)])
def test_baseline_sizes_through_tiling(product, oracle, cfg, tile):
    """BASELINE.json's full configurations (C2: 100 taxa x 1 M DNA sites, C3: 200 taxa x 1 M
    protein sites, C5: 50 taxa x 200 k codon sites) are out of the oracle's reach in
    seconds, but an alignment made of K copies of one tile is not: the oracle evaluates
    the tile, and size-independent properties carry that to the full size --
    lnL(full) = K * lnL(tile), per-site lnL and scaler counts are periodic in the tile,
    and the likelihood does not depend on the root edge."""
    S, R, ntips, N = pc.CONFIGS[cfg]
    K = N // tile
    t = pc.Tree(ntips, 42, 43)
    subst, freqs, alpha = {4: (pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841), 20: (*pc.protein_model(), 0.5),
                           61: (*pc.codon_model(), 0.5)}[S]          # build_instance's models
    with pc.build_instance(oracle, states=S, rate_cats=R, ntips=ntips, nsites=tile, coded=True, tree=t) as small:
        l_tile = pc.full_traversal(small)
        tile_codes = small.codes
    with pc.build_instance(product, states=S, rate_cats=R, ntips=ntips, nsites=tile, coded=True, tree=t) as ref:
        l_ref = pc.full_traversal(ref)
    # the north star itself at 61 states (|dlnL| < 1e-6 per site), far tighter below; each engine
    # with its own eigen-solver
    tol = 1e-6 * tile if S > 20 else max(REL_LNL * abs(l_tile), PER_SITE * tile)
    record("baseline_tile", config=int(cfg[1]), tile_sites=tile, dlnl_per_site=abs(l_ref - l_tile) / tile,
           lnl_per_site=l_tile / tile)
    print(f"\n[{cfg}] |dlnL| per site, HIP vs oracle on the {tile}-site tile: {abs(l_ref - l_tile) / tile:.3e}")
    assert abs(l_ref - l_tile) <= tol
    with pc.Instance(product, ntips, S, tile * K, R, attributes=pc.PLL_ATTRIB_PATTERN_TIP) as big:
        big.set_model(subst, freqs, product.gamma_cats(alpha, R))
        cmap = pc.state_charmap(S)
        for k in range(ntips):
            big.set_tip_states(k, cmap, (np.tile(tile_codes[k], K) + 48).tobytes())
        big.tree = t
        l_full = pc.full_traversal(big)
        assert abs(l_full - K * l_ref) <= 1e-10 * abs(l_full)
        assert abs(l_full - K * l_tile) <= K * tol
        _, persite = big.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b),
                                  t.root_matrix, persite=True)
        assert np.array_equal(persite.reshape(K, tile), np.broadcast_to(persite[:tile], (K, tile)))
        sc = big.get_scaler(t.scaler_of(t.root_a))
        assert np.array_equal(sc.reshape(K, tile), np.broadcast_to(sc[:tile], (K, tile)))
        # another root edge: same likelihood
        t2 = pc.Tree(ntips, 42, 43)
        t2.set_root_edge(t.nedges // 3)
        big.tree = t2
        l_rerooted = pc.full_traversal(big)
        assert abs(l_rerooted - l_full) <= 1e-9 * abs(l_full)
    if S == 61:
        return
    # ... and with PLL_ATTRIB_SITE_REPEATS: K copies of a tile are the extreme of repeats -- no node has more classes
    # than the tile has sites, so every operation that has a consumer is computed per class, level above level up to
    # the root edge.  Same likelihood, per-site values and scaler counts, bit for bit.
    with pc.Instance(product, ntips, S, tile * K, R, attributes=pc.PLL_ATTRIB_PATTERN_TIP | pc.PLL_ATTRIB_SITE_REPEATS) as rep:
        rep.set_model(subst, freqs, product.gamma_cats(alpha, R))
        cmap = pc.state_charmap(S)
        for k in range(ntips):
            rep.set_tip_states(k, cmap, (np.tile(tile_codes[k], K) + 48).tobytes())
        rep.tree = t
        assert pc.full_traversal(rep) == l_full
        _, persite_rep = rep.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b),
                                      t.root_matrix, persite=True)
        assert np.array_equal(persite_rep, persite)
        assert np.array_equal(rep.get_scaler(t.scaler_of(t.root_a)), sc)
        st = rep.repeat_stats()
        assert st.cherries >= ntips - 4 and st.classes * 100 < st.sites
        rep.tree = t2
        assert pc.full_traversal(rep) == l_rerooted


@pytest.mark.parametrize("states,coded,repeats", [(20, True, False), (4, False, False), (61, True, False), (10, False, False),
                                                  (2, True, False), (20, True, True), (4, True, True), (10, False, True)])
def test_host_mirrors_round_trip(product, states, coded, repeats):
    """what a checkpoint loader does on the GPU library (include/pllhip.h,
    PLLHIP_ATTRIB_HOST_MIRRORS): sync_to_host on the source sets the attribute; a partition
    created with it has host arrays a loader can fill (here: memmove, array by array as
    src/binary/binary_io_operations.c:194-314 walks them); sync_to_device moves them to the
    GPU, and the edge log-likelihood is there without recomputing anything."""
    kw = dict(states=states, rate_cats=4, ntips=10, nsites=301, coded=coded,
              attributes=pc.PLL_ATTRIB_SITE_REPEATS if repeats else 0)
    with pc.build_instance(product, **kw) as a:
        want = pc.full_traversal(a)
        assert product.lib.pllhip_sync_to_host(a.p, pc.PLLHIP_SYNC_ALL)
        pa = a.p.contents
        assert pa.attributes & pc.PLLHIP_ATTRIB_HOST_MIRRORS
        if repeats:
            # with the attribute the reference's walk goes through partition->repeats (src/binary/
            # binary_io_operations.c:231-236, 265-282): "no node is compressed", every vector for every site
            assert bool(pa.attributes & pc.PLL_ATTRIB_SITE_REPEATS) and bool(pa.repeats)
            rp = pa.repeats.contents
            assert all(rp.pernode_ids[i] == 0 for i in range(pa.nodes))
            assert all(rp.perscale_ids[i] == 0 for i in range(pa.scale_buffers))
            assert all(rp.pernode_allocated_clvs[i] == product.lib.pll_get_sites_number(a.p, i) == 301 for i in range(pa.nodes))
            assert all(not rp.pernode_site_id[i] and not rp.pernode_id_site[i] for i in range(pa.nodes))
            if states in (4, 20) and coded:
                assert a.repeat_stats().cherries > 0          # (the engine did compute per class of sites)
        else:
            assert not pa.repeats
        b = pc.Instance(product, 10, states, 301, 4, attributes=pa.attributes)
        with b:
            pb = b.p.contents
            S, Sp, R, N = a.S, a.Sp, a.R, a.N
            C.memmove(pb.eigen_decomp_valid, pa.eigen_decomp_valid, 4)
            for dst, src, n in ((pb.eigenvecs[0], pa.eigenvecs[0], S * Sp), (pb.inv_eigenvecs[0], pa.inv_eigenvecs[0], S * Sp),
                                (pb.eigenvals[0], pa.eigenvals[0], Sp), (pb.pmatrix[0], pa.pmatrix[0], pa.prob_matrices * R * S * Sp),
                                (pb.subst_params[0], pa.subst_params[0], S * (S - 1) // 2), (pb.frequencies[0], pa.frequencies[0], Sp),
                                (pb.rates, pa.rates, R), (pb.rate_weights, pa.rate_weights, R), (pb.prop_invar, pa.prop_invar, 1)):
                C.memmove(dst, src, 8 * n)
            first = 0
            if coded:
                for t in range(10):
                    C.memmove(pb.tipchars[t], pa.tipchars[t], N)
                C.memmove(pb.charmap, pa.charmap, 256)
                C.memmove(pb.tipmap, pa.tipmap, 256 * 8)
                pb.maxstates = pa.maxstates
                first = 10
            for i in range(first, pa.tips + pa.clv_buffers):
                C.memmove(pb.clv[i], pa.clv[i], 8 * N * R * Sp)
            for i in range(pa.scale_buffers):
                C.memmove(pb.scale_buffer[i], pa.scale_buffer[i], 4 * N)
            C.memmove(pb.pattern_weights, pa.pattern_weights, 4 * N)
            assert product.lib.pllhip_sync_to_device(b.p, pc.PLLHIP_SYNC_ALL)
            t = a.tree
            got = b.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
            assert got == want
            assert b.counters().partial_ops == 0


def test_rccl_reduce_callback_single_rank(product):
    """the native reduce callback (RCCL ncclAllReduce behind the reference's
    parallel_reduce_cb signature) on a 1-rank communicator: values come back
    unchanged for SUM / MAX / MIN; exercises id exchange, communicator set-up,
    staging buffers and stream synchronisation on the real device"""
    L = product.lib
    idbuf = C.create_string_buffer(128)
    assert L.pllhip_comm_get_unique_id(idbuf), product.errmsg
    comm = L.pllhip_comm_create(idbuf.raw, 0, 1, 0)
    assert comm, product.errmsg
    for op in (0, 1, 2):
        data = np.array([-1234.5678, 2.5, 0.0, 1e-300, -7e200])
        want = data.copy()
        L.pllhip_reduce_cb(comm, data.ctypes.data_as(pc.c_double_p), len(data), op)
        assert np.array_equal(data, want)
    big = np.arange(3000, dtype=np.float64)            # larger than one staging chunk
    L.pllhip_reduce_cb(comm, big.ctypes.data_as(pc.c_double_p), len(big), 0)
    assert np.array_equal(big, np.arange(3000, dtype=np.float64))
    L.pllhip_comm_destroy(comm)


def test_deferred_pmatrix_requests(product, oracle):
    """pll_update_prob_matrices calls are queued and launched in batches; a later
    request for the same matrix wins, a model change flushes the queue first, and
    the host mirror reflects every request"""
    a, b = _pair(product, oracle, states=20, rate_cats=4, ntips=6, nsites=64, coded=True)
    with a, b:
        for inst in (a, b):
            inst.update_pmatrices([0], [0.5])
            inst.update_pmatrices([0, 1], [0.25, 0.125])        # replaces matrix 0
            p = inst.p.contents
            for r in range(4):
                p.rates[r] = [0.3, 0.7, 1.0, 2.0][r]            # model change: matrix 2 uses new rates
            inst.update_pmatrices([2], [0.3])
        for m in range(3):
            assert np.allclose(a.get_pmatrix(m), b.get_pmatrix(m), rtol=1e-10, atol=1e-15)
        c = a.counters()
        assert c.pmatrix_updates == 4 and c.pmatrix_launches == 2


@pytest.mark.parametrize("states,ntips,rate_cats", [(4, 400, 4), (20, 200, 4), (61, 100, 4), (7, 250, 3), (2, 500, 4),
                                                    (16, 200, 2), (10, 220, 4), (20, 200, 3), (61, 100, 1), (33, 120, 2)])
def test_per_rate_scalers(product, oracle, states, ntips, rate_cats):
    """PLL_ATTRIB_RATE_SCALERS (one count per (site, rate), scaler[n*R + r]) on a deep tree with
    strong rate heterogeneity: counts bit-exact against the oracle, lnL / derivatives in
    tolerance, and the same likelihood as with per-site scalers"""
    # 61 states: with alpha = 0.3 the slowest category (rate 0.005) has P-matrix entries of 1e-19,
    # which NO fp64 eigen sum resolves (tests/test_expm_fixtures.py): its CLVs are noise in both
    # engines and their 2^-256 crossings are not comparable; alpha = 1 keeps every entry resolved
    kw = dict(states=states, rate_cats=rate_cats, ntips=ntips, nsites=131, coded=True,
              alpha=0.3 if states <= 20 else 1.0, attributes=pc.PLL_ATTRIB_RATE_SCALERS)
    a = pc.build_instance(product, **kw)
    b = pc.build_instance(oracle, **kw, tree=a.tree)
    kw.pop("attributes")
    c = pc.build_instance(product, **kw, tree=a.tree)
    with a, b, c:
        la, lb, lc = pc.full_traversal(a), pc.full_traversal(b), pc.full_traversal(c)
        assert lnl_close(la, lb, a.N, states), (la, lb)
        assert abs(la - lc) < 1e-9 * abs(la)
        t = a.tree
        seen_diff = False
        for op in t.ops:
            sa, sb = a.get_scaler(op[1]), b.get_scaler(op[1])
            assert sa.shape == (a.N * a.R,) and np.array_equal(sa, sb), f"scaler {op[1]}"
            seen_diff |= bool((sb.reshape(a.N, a.R).max(axis=1) != sb.reshape(a.N, a.R).min(axis=1)).any())
        assert seen_diff or rate_cats == 1, "the rates never differed in their counts: not a test of per-rate scaling"
        args = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        sta, stb = a.alloc_sumtable(), b.alloc_sumtable()
        a.update_sumtable(*args, sta)
        b.update_sumtable(*args, stb)
        for x in (0.02, 0.4):
            assert np.allclose(a.derivatives(args[2], args[3], x, sta), b.derivatives(args[2], args[3], x, stb),
                               rtol=1e-8 if states <= 20 else 2e-6)
        df, ddf = a.derivatives_multi(args[2], args[3], [0.02, 0.4, 1.5], sta)
        assert (df[1], ddf[1]) == a.derivatives(args[2], args[3], 0.4, sta)
        a.free_sumtable(sta)
        b.free_sumtable(stb)
        ra, rb = a.root_lnl(t.root_a, args[2]), b.root_lnl(t.root_a, args[2])
        assert abs(ra - rb) < 1e-10 * abs(rb) if states <= 20 else abs(ra - rb) < PER_SITE_61 * a.N
        assert product.lib.pllhip_sync_to_host(a.p, pc.PLLHIP_SYNC_SCALERS)        # mirrors hold sites x rates counts
        m = np.ctypeslib.as_array(a.p.contents.scale_buffer[t.scaler_of(t.root_a)], shape=(a.N * a.R,))
        assert np.array_equal(m, a.get_scaler(t.scaler_of(t.root_a)))


@pytest.mark.parametrize("asc_type", [pc.PLL_ATTRIB_AB_LEWIS, pc.PLL_ATTRIB_AB_FELSENSTEIN, pc.PLL_ATTRIB_AB_STAMATAKIS])
@pytest.mark.parametrize("states,coded,pinv", [(4, True, 0.0), (4, False, 0.0), (7, True, 0.0), (20, True, 0.1),
                                               (61, True, 0.0), (17, True, 0.0), (2, True, 0.0)])
def test_ascertainment_bias_correction(product, oracle, states, coded, pinv, asc_type):
    """PLL_ATTRIB_AB_*: the S constant patterns behind the alignment, the closed-form correction of
    lnL and of its derivatives.  Checked against the engine's own ordinary primitives and finite
    differences (tests/test_oracle_properties.py::check_asc_bias), and against the oracle."""
    from test_oracle_properties import check_asc_bias
    tol = 1e-9 if states <= 20 else 1e-7
    g = check_asc_bias(product, states, asc_type, coded, pinv, tol)
    c = check_asc_bias(oracle, states, asc_type, coded, pinv, tol)
    for x, y in zip(g, c):
        assert abs(x - y) <= (1e-9 if states <= 20 else 2e-6) * max(1.0, abs(y)), (g, c)


def test_ascertainment_bias_through_deferred_results(product):
    """an AB partition inside a result group (the evaluation driver on deferred results): the
    host-side correction still reaches the slot"""
    t = pc.Tree(8, 42, 43)
    ev = pc.Evaluation(product, t.newick(), nparts=2)
    with ev:
        for k, attrs in enumerate((pc.PLL_ATTRIB_AB_FLAG | pc.PLL_ATTRIB_AB_LEWIS, 0)):
            inst = pc.Instance(product, 8, 4, 300, 4, attributes=pc.PLL_ATTRIB_PATTERN_TIP | attrs)
            inst.set_model(pc.DNA_GTR_RATES, pc.DNA_FREQS, product.gamma_cats(0.8, 4))
            codes = pc.simulated_codes(t, 300, 4, seed=45 + k)
            cmap = pc.state_charmap(4)
            for tip in range(8):
                inst.set_tip_states(ev.tip_clv[tip], cmap, (codes[tip] + 48).tobytes())
            if attrs:
                inst.set_asc(pc.PLL_ATTRIB_AB_LEWIS)
            assert product.lib.pllhip_eval_set_partition(ev.ev, k, inst.p, inst.params_p)
            ev.parts.append(inst)
        plain = (ev.loglh(), ev.optimize_branches(1e-4, 10.0, 0.01, 4, -1))
    ev = pc.Evaluation(product, t.newick(), nparts=2)
    with ev:
        for k, attrs in enumerate((pc.PLL_ATTRIB_AB_FLAG | pc.PLL_ATTRIB_AB_LEWIS, 0)):
            inst = pc.Instance(product, 8, 4, 300, 4, attributes=pc.PLL_ATTRIB_PATTERN_TIP | attrs)
            inst.set_model(pc.DNA_GTR_RATES, pc.DNA_FREQS, product.gamma_cats(0.8, 4))
            codes = pc.simulated_codes(t, 300, 4, seed=45 + k)
            cmap = pc.state_charmap(4)
            for tip in range(8):
                inst.set_tip_states(ev.tip_clv[tip], cmap, (codes[tip] + 48).tobytes())
            if attrs:
                inst.set_asc(pc.PLL_ATTRIB_AB_LEWIS)
            assert product.lib.pllhip_eval_set_partition(ev.ev, k, inst.p, inst.params_p)
            ev.parts.append(inst)
        ev.attach_comm(None)
        fused = (ev.loglh(), ev.optimize_branches(1e-4, 10.0, 0.01, 4, -1))
    assert plain == fused
