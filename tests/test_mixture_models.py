"""Mixture models: heterogeneous per-rate-category parameter indices.

The reference carries one `param_indices[rate_cat]` array per partition (src/tree/treeinfo.c:288-306)
and hands it to every kernel call (src/optimize/pll_optimize.c:79, 192, 284-346, 805); LG4M / LG4X are
four-matrix mixtures (src/util/models_aa.c:57-67, src/util/pllmod_util.h:56-64, 99-122).  Rate category
r then takes its eigen-system, frequencies and proportion of invariant sites from rate matrix
`indices[r]`.

Pinned twice:
  * tests/golden/mixture_fixtures.npz (tests/golden/make_expm_fixtures.py mixtures): 60-digit matrix
    exponentials and their first / second derivatives from mpmath, brute-force pruning in numpy;
    neither engine's code took part.  Checked on the oracle (CPU) and the HIP engine (GPU): P-matrices,
    CLVs, per-site lnL, lnL, d(-lnL)/dt and d2(-lnL)/dt2 at the root edge.
  * HIP engine vs oracle on seeded instances with scaling (14 taxa, 1 031 sites): S in {4, 20, 61, 10}
    x index patterns {0,1,0,1}, {0,1,2,3}: P-matrices, CLVs, scalers (exact), lnL, root lnL, sumtable,
    derivatives (single and multi-length), per-branch P-matrix calls with alternating index sets,
    direct host writes to subst_params[1].
"""
import os

import numpy as np
import pytest

import common
import pllhip_ctypes as pc
from conftest import ROOT

FIX = os.path.join(ROOT, "tests", "golden", "mixture_fixtures.npz")
NONE = pc.PLL_SCALE_BUFFER_NONE
CASES = ["aa_mix2", "aa_mix4", "dna_mix2", "codon_mix2"]
TOL_P = 1e-14
TOL = {4: {"clv": 1e-12, "site": 1e-10, "lnl": 1e-10, "d": 1e-9},
       20: {"clv": 1e-11, "site": 1e-9, "lnl": 1e-9, "d": 1e-8},
       61: {"clv": 2e-6, "site": 1e-6, "lnl": 1e-7, "d": 2e-5}}


def run_fixture_case(lib, fx, name, coded):
    S = int(fx[f"{name}_states"])
    codes, gaps = fx[f"{name}_codes"], fx[f"{name}_gaps"]
    ntips, nsites = codes.shape
    idx = fx[f"{name}_indices"]
    R, nm = len(idx), int(fx[f"{name}_nmodels"])
    inst = pc.Instance(lib, ntips, S, nsites, R, attributes=pc.PLL_ATTRIB_PATTERN_TIP if coded else 0,
                       scalers=False, rate_matrices=nm)
    with inst:
        for m in range(nm):
            inst.set_model(fx[f"{name}_subst"][m], fx[f"{name}_freqs"][m], fx[f"{name}_rates"],
                           fx[f"{name}_weights"], idx=m)
        inst.set_params_indices(idx)
        cmap = pc.state_charmap(S)
        for t in range(ntips):
            seq = (codes[t] + 48).astype(np.uint8)
            seq[gaps[t]] = ord("-")
            inst.set_tip_states(t, cmap, seq.tobytes())
        if fx[f"{name}_pinv"].max() > 0:
            assert inst.L.pll_update_invariant_sites(inst.p)
            inv = np.ctypeslib.as_array(inst.p.contents.invariant, shape=(nsites,))
            assert np.array_equal(inv, fx[f"{name}_invariant"])
            for m in range(nm):
                inst.set_pinv(float(fx[f"{name}_pinv"][m]), idx=m)
        brl = fx["brlens"]
        # one call per branch, as the reference issues them (src/tree/treeinfo.c:845-865)
        inst.update_pmatrices(np.arange(len(brl)), brl, one_by_one=True)
        ops = [(p, NONE, c1, e1, NONE, c2, e2, NONE) for p, c1, e1, c2, e2 in fx["ops"]]
        inst.update_partials(ops)
        pa, ch, e = (int(x) for x in fx["root_edge"])
        lnl, persite = inst.edge_lnl(pa, NONE, ch, NONE, e, persite=True)
        Pw = fx[f"{name}_pmatrix"]
        dP = max(np.abs(inst.get_pmatrix(int(e_)) - Pw[k]).max() for k, e_ in enumerate(fx["keep_edges"]))
        dclv = 0.0
        for k, node in enumerate(int(x) for x in fx["keep_nodes"]):
            got, want = inst.get_clv(node), fx[f"{name}_clv_inner"][k]
            site_max = want.max(axis=(1, 2), keepdims=True)
            dclv = max(dclv, float((np.abs(got - want) / site_max).max()))
        dsite = float(np.abs(persite - fx[f"{name}_persite_lnl"]).max())
        dlnl = abs(lnl - float(fx[f"{name}_lnl"])) / nsites
        # derivatives at the root edge: one sumtable, then every fixture length singly and all in one call
        st = inst.alloc_sumtable()
        inst.update_sumtable(pa, ch, NONE, NONE, st)
        ts = fx[f"{name}_deriv_lengths"]
        dd = 0.0
        single = [inst.derivatives(NONE, NONE, float(t), st) for t in ts]
        for k in range(len(ts)):
            want = np.array([fx[f"{name}_df"][k], fx[f"{name}_ddf"][k]])
            dd = max(dd, float(np.max(np.abs(np.array(single[k]) - want) / np.maximum(1.0, np.abs(want)))))
        if hasattr(inst.L, "pllhip_compute_likelihood_derivatives_multi"):
            mdf, mddf = inst.derivatives_multi(NONE, NONE, ts, st)
            for k in range(len(ts)):
                assert np.allclose([mdf[k], mddf[k]], single[k], rtol=1e-10, atol=1e-9), (k, mdf, mddf, single)
        inst.free_sumtable(st)
    return {"dP": dP, "dCLV_rel_site_max": dclv, "max_persite_dlnl": dsite, "dlnl_per_site": dlnl, "dderiv_rel": dd}


def check_fixture(lib, which, name, coded):
    fx = np.load(FIX)
    d = run_fixture_case(lib, fx, name, coded)
    print(f"\n[{which}] {name} coded={coded}: " + "  ".join(f"{k}={v:.3e}" for k, v in d.items()))
    tol = TOL[int(fx[f"{name}_states"])]
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": "mixture_fixture", "engine": which, "case": name, "coded": coded,
                                **{k: float(v) for k, v in d.items()}}) + "\n")
    assert d["dP"] <= TOL_P, d
    assert d["dCLV_rel_site_max"] <= tol["clv"], d
    assert d["max_persite_dlnl"] <= tol["site"] and d["dlnl_per_site"] <= tol["lnl"], d
    assert d["dderiv_rel"] <= tol["d"], d


@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_oracle_against_mixture_fixtures(oracle, name, coded):
    check_fixture(oracle, "oracle", name, coded)


@pytest.mark.gpu
@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_hip_engine_against_mixture_fixtures(product, name, coded):
    check_fixture(product, "hip", name, coded)


# ---------------------------------------------------------------------------
# HIP engine vs oracle, seeded instances with scaling
# ---------------------------------------------------------------------------
PATTERNS = {"0101": [0, 1, 0, 1], "0123": [0, 1, 2, 3]}


def _lnl_tol(states, nsites, ref):
    return 5e-8 * nsites if states > 20 else max(1e-12 * abs(ref), 2e-9 * nsites)


def _mixture_pair(product, oracle, states, pattern, coded, attributes=0, with_pinv=True):
    ntips, nsites = (9, 257) if states > 20 else (14, 1031)
    idx = PATTERNS[pattern]
    pinv = [0.0, 0.12, 0.05, 0.2][:max(idx) + 1] if with_pinv and states <= 20 else None
    kw = dict(states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=coded, mixture=idx,
              mixture_pinv=pinv, attributes=attributes)
    a = pc.build_instance(product, **kw)
    b = pc.build_instance(oracle, **kw, tree=a.tree)
    return a, b


@pytest.mark.gpu
@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("pattern", ["0101", "0123"])
@pytest.mark.parametrize("states", [4, 20, 61, 10, 17])
def test_mixture_traversal_sumtable_derivatives(product, oracle, states, pattern, coded):
    a, b = _mixture_pair(product, oracle, states, pattern, coded)
    with a, b:
        assert a.p.contents.rate_matrices == max(PATTERNS[pattern]) + 1
        la, lb = pc.full_traversal(a, one_by_one_pmatrices=True), pc.full_traversal(b, one_by_one_pmatrices=True)
        assert np.isfinite(lb) and lb < 0
        assert abs(la - lb) <= _lnl_tol(states, a.N, lb), (la, lb)
        t = a.tree
        for m in (0, t.nedges // 2, t.nedges - 1):
            assert np.abs(a.get_pmatrix(m) - b.get_pmatrix(m)).max() < 1e-13
        # the categories really differ: rows of P under matrix 0 and matrix 1 at the same rate index differ
        for op in t.ops:
            ca, cb = a.get_clv(op[0]), b.get_clv(op[0])
            err = common.vec_err(ca, cb) if states <= 20 else \
                float(np.max(np.abs(ca - cb) / np.maximum(np.abs(cb).max(axis=(1, 2), keepdims=True), 1e-300)))
            assert err < (1e-8 if states <= 20 else 1e-7), (op[0], err)
            assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        ra, rb = a.root_lnl(t.root_a, sa), b.root_lnl(t.root_a, sa)
        assert abs(ra - rb) <= _lnl_tol(states, a.N, rb)
        _, pa_ = a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)
        _, pb_ = b.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)
        assert np.abs(pa_ - pb_).max() < (1e-9 if states <= 20 else 1e-6)
        sta, stb = a.alloc_sumtable(), b.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, sa, sb, sta)
        b.update_sumtable(t.root_a, t.root_b, sa, sb, stb)
        lens = [1e-4, 0.013, 0.1, 0.77, 5.0]
        rtol = 1e-9 if states <= 20 else 1e-5   # 61 states: each engine runs its own eigen-solver (see test_gpu_parity.py)
        for bl in lens:
            da, db = a.derivatives(sa, sb, bl, sta), b.derivatives(sa, sb, bl, stb)
            assert np.allclose(da, db, rtol=rtol, atol=1e-9 * a.N), (bl, da, db)
        mdf, mddf = a.derivatives_multi(sa, sb, lens, sta)
        for k, bl in enumerate(lens):
            db = b.derivatives(sa, sb, bl, stb)
            assert np.allclose([mdf[k], mddf[k]], db, rtol=rtol, atol=1e-9 * a.N), (bl, mdf[k], mddf[k], db)
        a.free_sumtable(sta); b.free_sumtable(stb)


@pytest.mark.gpu
@pytest.mark.parametrize("states", [4, 20, 61])
def test_mixture_with_per_rate_scalers(product, oracle, states):
    """PLL_ATTRIB_RATE_SCALERS together with a mixture: the rate's own count, the rate's own model"""
    a, b = _mixture_pair(product, oracle, states, "0101", True, attributes=pc.PLL_ATTRIB_RATE_SCALERS, with_pinv=False)
    with a, b:
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la - lb) <= _lnl_tol(states, a.N, lb), (la, lb)
        for op in a.tree.ops:
            assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("states", [4, 20, 61, 10])
def test_alternating_index_sets_in_consecutive_pmatrix_calls(product, oracle, states):
    """consecutive per-branch pll_update_prob_matrices calls with DIFFERENT params_indices: the
    deferred request queue has to launch what it holds before it takes a request made under another
    index set (pll_core.hip, pll_update_prob_matrices), and a re-request of a queued matrix under
    another set replaces it"""
    a, b = _mixture_pair(product, oracle, states, "0101", True, with_pinv=False)
    with a, b:
        t = a.tree
        sets = [pc._u32([0, 1, 0, 1]), pc._u32([1, 0, 1, 0]), pc._u32([1, 1, 1, 1]), pc._u32([0, 0, 0, 0])]
        for inst in (a, b):
            for k in range(t.nedges):
                inst.params = sets[k % len(sets)]
                inst.update_pmatrices([k], [t.brlens[k]])
            # matrix 0 once more under another set before anything consumed the first request
            inst.params = sets[2]
            inst.update_pmatrices([0], [t.brlens[0] * 2.0])
            inst.params = sets[0]
        for m in range(t.nedges):
            assert np.abs(a.get_pmatrix(m) - b.get_pmatrix(m)).max() < 1e-13, m
        # matrix m was built under sets[m % 4]: under set 2 every category uses model 1, under 3 model 0
        P2, P3 = a.get_pmatrix(2), a.get_pmatrix(3)
        assert np.abs(P2 - P3).max() > 1e-4
        for inst in (a, b):
            inst.update_partials(inst.make_ops(t.ops_with_scalers(True)), len(t.ops))
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        la = a.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
        lb = b.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
        assert abs(la - lb) <= _lnl_tol(states, a.N, lb), (la, lb)


@pytest.mark.gpu
@pytest.mark.parametrize("states", [4, 20])
def test_direct_host_write_to_second_rate_matrix(product, oracle, states):
    """pll-modules writes partition->subst_params[i] / frequencies[i] for i > 0 without a setter and
    clears eigen_decomp_valid[i] (src/algorithm/algo_callback.c:44-68): the engine notices"""
    a, b = _mixture_pair(product, oracle, states, "0101", True, with_pinv=False)
    with a, b:
        l0 = pc.full_traversal(a)
        assert abs(l0 - pc.full_traversal(b)) <= _lnl_tol(states, a.N, l0)
        for inst in (a, b):
            p = inst.p.contents
            p.subst_params[1][0] *= 3.0
            p.subst_params[1][2] *= 0.4
            f = np.ctypeslib.as_array(p.frequencies[1], shape=(states,))
            f[0], f[1] = f[0] + 0.5 * f[1], 0.5 * f[1]
            p.eigen_decomp_valid[1] = 0
        la, lb = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la - l0) > 1e-3
        assert abs(la - lb) <= _lnl_tol(states, a.N, lb), (la, lb)
        # matrix 0 untouched: a partition that uses matrix 0 for every category sees no change
        for inst in (a, b):
            inst.set_params_indices([0, 0, 0, 0])
        la0, lb0 = pc.full_traversal(a), pc.full_traversal(b)
        assert abs(la0 - lb0) <= _lnl_tol(states, a.N, lb0)


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True])
def test_parameter_change_between_two_pmatrix_calls(product, oracle, sharded):
    """pll_update_prob_matrices -> pll_set_subst_params (+ pll_update_eigen) -> pll_update_prob_matrices
    with no other kernel entry in between: the second call must build its matrices from the NEW
    eigen-system (the light model check of a P-matrix burst skipped the eigen compare)"""
    if sharded:
        dev = (pc.C.c_int * 2)(0, 0)
        assert product.lib.pllhip_set_sharding(2, dev)
    try:
        a = pc.build_instance(product, states=20, rate_cats=4, ntips=8, nsites=700, coded=True)
    finally:
        if sharded:
            product.lib.pllhip_set_sharding(0, None)
    b = pc.build_instance(oracle, states=20, rate_cats=4, ntips=8, nsites=700, coded=True, tree=a.tree)
    with a, b:
        if sharded:
            assert product.lib.pllhip_shard_count(a.p) == 2
        t = a.tree
        sub2 = pc._f64(np.asarray(pc.protein_model()[0]) * (1.0 + 0.5 * pc.uniform01(77, 190)))
        for explicit_eigen in (True, False):
            for inst in (a, b):
                inst.update_pmatrices(np.arange(t.nedges), t.brlens)
                inst.L.pll_set_subst_params(inst.p, 0, sub2.ctypes.data_as(pc.c_double_p))
                if explicit_eigen:
                    assert inst.L.pll_update_eigen(inst.p, 0)
                inst.update_pmatrices(np.arange(t.nedges), t.brlens)
            for m in (0, t.nedges - 1):
                assert np.abs(a.get_pmatrix(m) - b.get_pmatrix(m)).max() < 1e-13, (explicit_eigen, m)
            sub2 = pc._f64(sub2[::-1].copy())
