"""Mathematical invariants of the CPU oracle for everything the reference's golden
files cannot pin offline (scaling, p-inv, 20/61 states, more than 3 taxa, pattern
weights, tip codes): SURVEY.md section 7 step 2.  The HIP engine is then compared
with the oracle (tests/test_gpu_parity.py), so these properties carry over."""
import numpy as np
import pytest

import pllhip_ctypes as pc
from conftest import ORACLE_LIB

NONE = pc.PLL_SCALE_BUFFER_NONE


@pytest.mark.parametrize("states", [4, 5, 20])
def test_rerooting_invariance(oracle, states):
    with pc.build_instance(oracle, states=states, rate_cats=4, ntips=9, nsites=60, coded=True) as a:
        ref = pc.full_traversal(a)
        for k in range(a.tree.nedges):
            a.tree.set_root_edge(k)
            assert abs(pc.full_traversal(a) - ref) < 1e-10 * abs(ref)


@pytest.mark.parametrize("states,ntips", [(4, 120), (20, 70)])
def test_scaling_on_equals_scaling_off_before_underflow(oracle, states, ntips):
    kw = dict(states=states, rate_cats=4, ntips=ntips, nsites=25, coded=True)
    with pc.build_instance(oracle, scalers=True, **kw) as a, pc.build_instance(oracle, scalers=False, **kw) as b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert np.isfinite(lb) and abs(la - lb) < 1e-9 * abs(la)


def test_scaling_keeps_deep_trees_finite(oracle):
    kw = dict(states=4, rate_cats=4, ntips=700, nsites=7, coded=True)
    with pc.build_instance(oracle, scalers=True, **kw) as a, pc.build_instance(oracle, scalers=False, **kw) as b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert np.isfinite(la) and la < 0
        assert not np.isfinite(lb)                         # without scalers the product underflows
        assert a.get_scaler(a.tree.scaler_of(a.tree.root_a)).max() >= 1


@pytest.mark.parametrize("states", [4, 20])
def test_tip_codes_equal_full_tip_clvs(oracle, states):
    kw = dict(states=states, rate_cats=4, ntips=8, nsites=90)
    with pc.build_instance(oracle, coded=True, **kw) as a, pc.build_instance(oracle, coded=False, **kw) as b:
        assert abs(pc.full_traversal(a) - pc.full_traversal(b)) < 1e-12 * 1e4
        for op in a.tree.ops:
            assert np.array_equal(a.get_clv(op[0]), b.get_clv(op[0]))


@pytest.mark.parametrize("states", [4, 20, 61])
def test_derivatives_match_finite_differences(oracle, states):
    with pc.build_instance(oracle, states=states, rate_cats=4, ntips=6, nsites=40, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        st = a.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, sa, sb, st)

        def neg_lnl(bl):
            a.update_pmatrices([t.root_matrix], [bl])
            return -a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
        for x in (0.05, 0.3, 1.5):
            h = 1e-5
            df, ddf = a.derivatives(sa, sb, x, st)
            f0, fp, fm = neg_lnl(x), neg_lnl(x + h), neg_lnl(x - h)
            assert abs(df - (fp - fm) / (2 * h)) < 1e-5 * max(1.0, abs(df))
            assert abs(ddf - (fp - 2 * f0 + fm) / (h * h)) < 5e-3 * max(1.0, abs(ddf))
        a.free_sumtable(st)


def test_pattern_weights_are_linear(oracle):
    """lnL with weights w equals the w-weighted sum of per-site lnL; duplicating a
    column equals giving it weight 2"""
    with pc.build_instance(oracle, states=4, rate_cats=4, ntips=7, nsites=50, coded=True) as a:
        t = a.tree
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        base, per = (pc.full_traversal(a), None)
        _, per = a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)
        assert abs(per.sum() - base) < 1e-10 * abs(base)
        w = (pc.splitmix64(3, 50) % np.uint64(4)).astype(np.uint32)
        a.set_pattern_weights(w)
        assert abs(a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix) - float(np.dot(per, w))) < 1e-9 * abs(base)
        assert a.p.contents.pattern_weight_sum == int(w.sum())


def test_invariant_sites_model(oracle):
    """p-inv: detection of invariant columns, pinv -> 0 limit, and the mixture formula
    L = (1-p) L_var(rates/(1-p)) + p * pi[state] on invariant columns"""
    with pc.build_instance(oracle, states=4, rate_cats=4, ntips=5, nsites=600, coded=True) as a:
        assert oracle.lib.pll_update_invariant_sites(a.p)
        inv = np.ctypeslib.as_array(a.p.contents.invariant, shape=(a.N,)).copy()
        codes = a.codes
        same = (codes == codes[0]).all(axis=0)
        assert np.array_equal(inv >= 0, same)
        assert np.array_equal(inv[same], codes[0][same].astype(np.int32))
        assert oracle.lib.pll_count_invariant_sites(a.p, None) == int(same.sum())
        l0 = pc.full_traversal(a)
        a.set_pinv(1e-9)
        assert abs(pc.full_traversal(a) - l0) < 1e-5
        a.set_pinv(0.3)
        t = a.tree
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        pc.full_traversal(a)
        _, per = a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)
        # variable part alone: same rates/(1-p) trick, no invariant term
        b = pc.build_instance(oracle, states=4, rate_cats=4, ntips=5, nsites=600, coded=True, tree=a.tree)
        with b:
            rates = np.ctypeslib.as_array(b.p.contents.rates, shape=(4,))
            rates /= 0.7
            pc.full_traversal(b)
            _, pvar = b.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)
        freqs = np.array(pc.DNA_FREQS)
        expect = np.log(0.7 * np.exp(pvar) + np.where(inv >= 0, 0.3 * freqs[np.maximum(inv, 0)], 0.0))
        assert np.allclose(per, expect, rtol=1e-10)


def test_identity_matrix_at_zero_branch_length(oracle):
    with pc.Instance(oracle, 3, 20, 4, 4, scalers=False, clv_buffers=1, prob_matrices=3) as a:
        r, f = pc.protein_model()
        a.set_model(r, f, oracle.gamma_cats(0.5, 4))
        a.update_pmatrices([0, 1], [0.0, 1e3])
        p0, pinf = a.get_pmatrix(0), a.get_pmatrix(1)
        assert np.array_equal(p0[0], np.eye(20))
        assert np.allclose(pinf[3], np.tile(f / np.sum(f), (20, 1)), atol=1e-9)   # stationary limit
        assert np.allclose(pinf.sum(axis=2), 1.0, atol=1e-12)


def test_tiled_alignment_scales_linearly(oracle):
    with pc.build_instance(oracle, states=20, rate_cats=4, ntips=6, nsites=40, coded=True) as base:
        l1 = pc.full_traversal(base)
        with pc.Instance(oracle, 6, 20, 40 * 7, 4, attributes=pc.PLL_ATTRIB_PATTERN_TIP) as big:
            r, f = pc.protein_model()
            big.set_model(r, f, oracle.gamma_cats(0.5, 4))
            cmap = pc.state_charmap(20)
            for t in range(6):
                big.set_tip_states(t, cmap, (np.tile(base.codes[t], 7) + 48).tobytes())
            big.tree = base.tree
            assert abs(pc.full_traversal(big) - 7 * l1) < 1e-10 * abs(l1) * 7


def test_error_reporting(oracle):
    L = oracle.lib
    assert not L.pll_partition_create(3, 1, 1, 4, 1, 3, 4, 0, 0)             # one state
    assert oracle.errno == 113
    with pc.Instance(oracle, 3, 4, 4, 4, scalers=False, clv_buffers=1, prob_matrices=3) as a:
        with pytest.raises(RuntimeError):
            a.set_tip_states(0, pc.state_charmap(4), b"01Z3")                   # illegal character
        assert oracle.errno == 114
        with pytest.raises(RuntimeError):
            a.set_pinv(1.5)
        assert oracle.errno == 118


def test_vectorised_partials_agree_with_the_scalar_path():
    """ORC_FAST=1 (what bench.py's cpu_baseline times) reorders the sums of the 4-, 20- and
    61-state partials; it must give the scalar path's numbers to rounding, scalers exactly"""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import pllhip_ctypes as pc
lib = pc.PllLib(%r)
for S, n in ((20, 120), (4, 300), (61, 60)):
    with pc.build_instance(lib, states=S, rate_cats=4, ntips=n, nsites=61, coded=True) as a:
        l = pc.full_traversal(a)
        root = a.tree.root_a
        print(repr(l), a.get_scaler(a.tree.scaler_of(root)).tolist(), repr(float(np.abs(a.get_clv(root)).sum())))
""" % (os.path.dirname(pc.__file__), ORACLE_LIB)
    out = []
    for fast in ("0", "1"):
        env = dict(os.environ, ORC_FAST=fast)
        out.append(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True,
                                  text=True, timeout=300).stdout.splitlines())
    for a, b in zip(*out):
        la, sa, ca = a.split(" ", 1)[0], a[a.index("["):a.index("]") + 1], a.rsplit(" ", 1)[1]
        lb, sb, cb = b.split(" ", 1)[0], b[b.index("["):b.index("]") + 1], b.rsplit(" ", 1)[1]
        assert sa == sb and any(ch in "123456789" for ch in sa)    # same scaling decisions; scaling happened
        assert abs(float(la) - float(lb)) < 1e-12 * abs(float(la))
        assert abs(float(ca) - float(cb)) < 1e-11 * abs(float(ca))


@pytest.mark.parametrize("states,ntips", [(4, 300), (20, 150), (7, 200)])
def test_per_rate_scalers_agree_with_per_site_scalers(oracle, states, ntips):
    """PLL_ATTRIB_RATE_SCALERS: one count per (site, rate).  On a deep tree with strong rate
    heterogeneity the slow and the fast categories cross 2^-256 at different depths; the
    likelihood must not depend on which scaling scheme carried it there."""
    kw = dict(states=states, rate_cats=4, ntips=ntips, nsites=61, coded=True, alpha=0.3)
    with pc.build_instance(oracle, **kw) as a, \
         pc.build_instance(oracle, **kw, attributes=pc.PLL_ATTRIB_RATE_SCALERS) as b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert np.isfinite(la) and abs(la - lb) < 1e-9 * abs(la)
        t = a.tree
        sa = a.get_scaler(t.scaler_of(t.root_a))
        sb = b.get_scaler(t.scaler_of(t.root_a)).reshape(a.N, a.R)
        assert sa.max() >= 1 and (sb.max(axis=1) != sb.min(axis=1)).any()   # the rates really differ
        # a per-site count can never exceed any of the per-rate counts of its site
        assert (sa[:, None] <= sb).all()
        # derivatives: same function, same slope and curvature
        sta, stb = a.alloc_sumtable(), b.alloc_sumtable()
        args = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        a.update_sumtable(*args, sta)
        b.update_sumtable(*args, stb)
        for x in (0.01, 0.3):
            da, db = a.derivatives(args[2], args[3], x, sta), b.derivatives(args[2], args[3], x, stb)
            assert np.allclose(da, db, rtol=1e-8)
        a.free_sumtable(sta)
        b.free_sumtable(stb)
        # root form
        ra = a.root_lnl(t.root_a, t.scaler_of(t.root_a))
        rb = b.root_lnl(t.root_a, t.scaler_of(t.root_a))
        assert abs(ra - rb) < 1e-9 * abs(ra)


def check_asc_bias(lib, states, asc_type, coded, pinv=0.0, tol=1e-9):
    """ascertainment-bias correction against the same library's ordinary primitives: the
    corrected lnL must equal lnL(alignment) + f(likelihoods of the S constant patterns), the
    latter taken from an ordinary partition that holds just those S columns; derivatives
    against finite differences of the corrected lnL"""
    ntips, nsites, R = 9, 150, 4
    tree = pc.Tree(ntips, 42, 43)
    ab = pc.build_instance(lib, states=states, rate_cats=R, ntips=ntips, nsites=nsites, coded=coded, tree=tree,
                           attributes=pc.PLL_ATTRIB_AB_FLAG | asc_type, pinv=pinv)
    plain = pc.build_instance(lib, states=states, rate_cats=R, ntips=ntips, nsites=nsites, coded=coded, tree=tree,
                              pinv=pinv)
    const = pc.build_instance(lib, states=states, rate_cats=R, ntips=ntips, nsites=states, coded=coded, tree=tree,
                              pinv=pinv)
    with ab, plain, const:
        rng = np.random.RandomState(3)
        w = rng.randint(1, 5, size=nsites).astype(np.uint32)
        sw = rng.randint(0, 7, size=states).astype(np.uint32)
        sw[0] = max(1, sw[0])
        for inst in (ab, plain):
            inst.set_pattern_weights(w)
        ab.set_asc(asc_type, sw)
        cmap = pc.state_charmap(states)
        for t in range(ntips):
            const.set_tip_states(t, cmap, (np.arange(states, dtype=np.uint8) + 48).tobytes())
        if pinv > 0:
            for inst in (ab, plain, const):
                assert inst.L.pll_update_invariant_sites(inst.p)
        l_ab, l_plain = pc.full_traversal(ab), pc.full_traversal(plain)
        pc.full_traversal(const)
        sa, sb = tree.scaler_of(tree.root_a), tree.scaler_of(tree.root_b)
        _, lk = const.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix, persite=True)
        W = float(w.sum())
        want = {pc.PLL_ATTRIB_AB_LEWIS: -W * np.log1p(-np.exp(lk).sum()),
                pc.PLL_ATTRIB_AB_FELSENSTEIN: float(sw.sum()) * np.log(np.exp(lk).sum()),
                pc.PLL_ATTRIB_AB_STAMATAKIS: float((sw * lk).sum())}[asc_type]
        assert abs(l_ab - (l_plain + want)) <= tol * abs(l_ab), (l_ab, l_plain, want)
        assert abs(want) > 1e-3                          # the correction is not a no-op
        # per-site values are those of the alignment patterns
        _, ps_ab = ab.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix, persite=True)
        _, ps_pl = plain.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix, persite=True)
        assert np.allclose(ps_ab, ps_pl, rtol=1e-12, atol=0)
        # derivatives of the corrected -lnL
        st = ab.alloc_sumtable()
        ab.update_sumtable(tree.root_a, tree.root_b, sa, sb, st)
        x, h = float(tree.brlens[tree.root_matrix]), 1e-5

        def neg_lnl(bl):
            ab.update_pmatrices([tree.root_matrix], [bl])
            return -ab.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix)
        df, ddf = ab.derivatives(sa, sb, x, st)
        f0, fp, fm = neg_lnl(x), neg_lnl(x + h), neg_lnl(x - h)
        assert abs(df - (fp - fm) / (2 * h)) < 1e-5 * max(1.0, abs(df)), (df, (fp - fm) / (2 * h))
        assert abs(ddf - (fp - 2 * f0 + fm) / (h * h)) < 5e-3 * max(1.0, abs(ddf))
        ab.free_sumtable(st)
        return l_ab, df, ddf


@pytest.mark.parametrize("asc_type", [pc.PLL_ATTRIB_AB_LEWIS, pc.PLL_ATTRIB_AB_FELSENSTEIN, pc.PLL_ATTRIB_AB_STAMATAKIS])
@pytest.mark.parametrize("states,coded,pinv", [(4, True, 0.0), (4, False, 0.0), (7, True, 0.0), (20, True, 0.1)])
def test_ascertainment_bias_correction_on_oracle(oracle, states, coded, pinv, asc_type):
    check_asc_bias(oracle, states, asc_type, coded, pinv)


def test_asc_needs_a_partition_created_for_it(oracle):
    with pc.Instance(oracle, 3, 4, 8, 4) as a:
        assert not oracle.lib.pll_set_asc_bias_type(a.p, pc.PLL_ATTRIB_AB_LEWIS)
        assert oracle.errno == 121          # PLL_ERROR_AB_INVALIDMETHOD
