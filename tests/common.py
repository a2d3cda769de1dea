"""Shared helpers of the test-suite: golden fixtures, a host-side replay of the
reference's single-partition Newton-Raphson branch-length optimisation, and
comparison utilities.  Everything here drives a library through the C ABI only.
"""
import json
import math
import os

import numpy as np

import pllhip_ctypes as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NONE = pc.PLL_SCALE_BUFFER_NONE
# tests/test_00_forced_modes.py runs the whole -m gpu suite with every partition treated as if it had
# PLL_ATTRIB_SITE_REPEATS (PLLHIP_SITE_REPEATS=2) and with evaluate-only traversals everywhere (PLLHIP_TRANSIENT=1):
# every NUMBER must stay what it is; asserts about how many launches / class operations a call took are void there
FORCED_REPEATS = os.environ.get("PLLHIP_SITE_REPEATS") == "2"
FORCED_TRANSIENT = os.environ.get("PLLHIP_TRANSIENT", "0") not in ("", "0")
FORCED = FORCED_REPEATS or FORCED_TRANSIENT


def fixtures():
    with open(os.path.join(GOLDEN, "blopt_fixtures.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------------------
# tree records in the shape of pll_unode_t (ring of three per inner node)
# ---------------------------------------------------------------------------
class Rec:
    __slots__ = ("next", "back", "clv", "scaler", "pmatrix", "length", "label")

    def __init__(self, clv, scaler, label=None):
        self.next = None
        self.back = None
        self.clv, self.scaler, self.label = clv, scaler, label
        self.pmatrix, self.length = 0, 0.0


def link(a, b, pmatrix, length):
    a.back, b.back = b, a
    a.pmatrix = b.pmatrix = pmatrix
    a.length = b.length = length


def records_from_tree(tree, use_scalers=True):
    """pc.Tree -> dict node -> list of records (1 for tips, 3 for inner)"""
    recs = {}
    for nd in tree.adj:
        sc = tree.scaler_of(nd) if use_scalers else NONE
        if nd < tree.ntips:
            recs[nd] = [Rec(nd, NONE, f"t{nd}")]
        else:
            ring = [Rec(nd, sc) for _ in range(3)]
            for k in range(3):
                ring[k].next = ring[(k + 1) % 3]
            recs[nd] = ring
    used = {nd: 0 for nd in recs}
    for k, (u, v) in enumerate(tree.edges):
        a = recs[u][used[u]]
        b = recs[v][used[v]]
        used[u] += 1
        used[v] += 1
        link(a, b, k, float(tree.brlens[k]))
    return recs


def three_taxon_records(brlens):
    """the tree built by test/src/optimize/blopt-minimal.c:123-139: tips = CLV
    0,1,2 on P-matrices 0,1,2; inner CLV 3; no scalers"""
    tips = [Rec(i, NONE, f"TIP {i + 1}") for i in range(3)]
    ring = [Rec(3, NONE) for _ in range(3)]
    for k in range(3):
        ring[k].next = ring[(k + 1) % 3]
        link(tips[k], ring[k], k, brlens[k])
    return tips, ring


# ---------------------------------------------------------------------------
# Newton-Raphson branch-length optimisation, replayed on the host.
# Behavioural restatement of the reference's single-partition path
#   pllmod_opt_optimize_branch_lengths_local   src/optimize/pll_optimize.c:961-1097
#   recomp_iterative                           src/optimize/pll_optimize.c:778-926
#   pllmod_opt_minimize_newton_old             src/optimize/opt_algorithms.c:281-384
# (method NEWTON_OLDFAST: no per-branch lnL check, tolerance = min_brlen / 10,
# 30 iterations).  Used to pin pll_update_sumtable +
# pll_compute_likelihood_derivatives against the golden post-BLO numbers.
# ---------------------------------------------------------------------------
class NewtonError(RuntimeError):
    pass


def newton_old(x1, xguess, x2, tol, max_iters, deriv):
    rts = min(max(xguess, x1), x2)
    f, df = deriv(rts)
    if not (math.isfinite(f) and math.isfinite(df)):
        raise NewtonError("wrong likelihood derivatives")
    if df >= 0.0 and abs(f) < tol:
        return rts
    xl, xh = (rts, x2) if f < 0.0 else (x1, rts)
    dx = abs(xh - xl)
    rts_old = rts
    for i in range(1, max_iters + 1):
        rts_old = rts
        if df <= 0.0 or ((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0:
            dx = 0.5 * (xh - xl)       # concave or out of bracket: bisect
            rts = xl + dx
            if xl == rts:
                return rts
        else:
            dx = f / df
            tmp = rts
            rts -= dx
            if tmp == rts:
                return rts
        if abs(dx) < tol:
            return rts_old
        if i == max_iters:
            break
        rts = max(rts, x1)
        f, df = deriv(rts)
        if not (math.isfinite(f) and math.isfinite(df)):
            raise NewtonError("wrong likelihood derivatives [it]")
        if df > 0.0 and abs(f) < tol:
            return rts
        if f < 0.0:
            xl = rts
        else:
            xh = rts
    raise NewtonError("exceeded maximum number of iterations")


def _one_op(inst, parent, c1, c2):
    inst.update_partials([(parent.clv, parent.scaler, c1.back.clv, c1.back.pmatrix, c1.back.scaler,
                           c2.back.clv, c2.back.pmatrix, c2.back.scaler)])


def _optimise_around(inst, st, p, radius, bmin, bmax, tol, trace):
    q = p.next
    z = q.next if q is not None else None
    xorig = p.length
    inst.update_sumtable(p.clv, p.back.clv, p.scaler, p.back.scaler, st)
    xguess = p.length if bmin <= p.length <= bmax else 0.1
    xres = newton_old(bmin, xguess, bmax, tol, 30,
                      lambda t: inst.derivatives(p.scaler, p.back.scaler, t, st))
    p.length = p.back.length = xres
    if trace is not None:
        trace.append((p.pmatrix, xres))
    if abs(xres - xorig) > 1e-10:
        inst.update_pmatrices([p.pmatrix], [xres])
    if radius and q is not None and z is not None:
        _one_op(inst, q, p, z)
        _optimise_around(inst, st, q.back, radius - 1, bmin, bmax, tol, trace)
        _one_op(inst, z, q, p)
        _optimise_around(inst, st, z.back, radius - 1, bmin, bmax, tol, trace)
        _one_op(inst, p, z, q)


def optimise_branch_lengths_local(inst, root, bmin, bmax, tolerance, smoothings, radius, trace=None):
    """returns -lnL like the reference; CLVs must be oriented towards `root`"""
    def edge():
        return inst.edge_lnl(root.back.clv, root.back.scaler, root.clv, root.scaler, root.pmatrix)

    lnl = edge()
    tol_nr = bmin / 10.0
    st = inst.alloc_sumtable()
    try:
        iters = smoothings
        while iters:
            _optimise_around(inst, st, root, radius, bmin, bmax, tol_nr, trace)
            if radius:
                _optimise_around(inst, st, root.back, radius - 1, bmin, bmax, tol_nr, trace)
            new = edge()
            if new - lnl > new * 1e-13:
                iters -= 1
                if abs(new - lnl) < tolerance:
                    iters = 0
                lnl = new
            else:
                lnl = new
                break
    finally:
        inst.free_sumtable(st)
    return -lnl


# ---------------------------------------------------------------------------
# the two golden cases, runnable against any library
# ---------------------------------------------------------------------------
def run_golden_case(lib, name, coded=False):
    fx = fixtures()[name]
    S, R, N = fx["states"], fx["rate_cats"], fx["sites"]
    attrs = pc.PLL_ATTRIB_PATTERN_TIP if coded else 0
    inst = pc.Instance(lib, 3, S, N, R, attributes=attrs, scalers=False, clv_buffers=1, prob_matrices=3)
    rates = lib.gamma_cats(fx["alpha"], R)
    inst.set_model(fx["subst_params"], fx["frequencies"], rates)
    if "tip_clv" in fx:
        for t in range(3):
            inst.set_tip_clv(t, fx["tip_clv"][t])
    else:
        for t in range(3):
            inst.set_tip_states(t, fx["charmap"], fx["sequences"][t].encode())
    bl = list(fx["branch_lengths"])
    inst.update_pmatrices([0, 1, 2], bl)
    out = {"pmatrix": [inst.get_pmatrix(m) for m in range(3)]}
    inst.update_partials([(3, NONE, 0, 0, NONE, 1, 1, NONE)])
    out["lnl_initial"] = inst.edge_lnl(3, NONE, 2, NONE, 2)
    tips, ring = three_taxon_records(bl)
    b = fx["blo"]
    out["neg_lnl_returned"] = optimise_branch_lengths_local(
        inst, tips[2].back, b["min"], b["max"], b["tolerance"], b["smoothings"], b["radius"])
    out["lnl_after_blo"] = inst.edge_lnl(3, NONE, 2, NONE, 2)
    out["brlens_after_blo"] = [tips[k].length for k in range(3)]
    inst.close()
    return fx["expected"], out


def check_golden(expected, got):
    for m in range(3):
        exp = np.array(expected["pmatrices"][m]["P"])
        assert np.allclose(np.round(got["pmatrix"][m], 4), exp, atol=1e-9), f"P-matrix {m}"
    # printed precision of the reference's golden files: %.10f lnL, %.6f brlen
    assert abs(got["lnl_initial"] - expected["lnl_initial"]) < 6e-11
    assert abs(got["lnl_after_blo"] - expected["lnl_after_blo"]) < 6e-11
    assert abs(got["neg_lnl_returned"] - expected["neg_lnl_returned"]) < 6e-11
    for g, e in zip(got["brlens_after_blo"], expected["brlens_after_blo"]):
        assert abs(g - e) < 6e-7


# ---------------------------------------------------------------------------
def vec_err(a, b):
    """largest deviation of a CLV-shaped array [..., states], measured against the
    largest entry of the same (site, rate) vector.  Entries many orders below their
    vector's maximum (e.g. multi-step codon changes, ~1e-13 of the row) come out of
    cancelling eigen sums and carry no weight in any likelihood."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    scale = np.maximum(np.abs(b).max(axis=-1, keepdims=True), 1e-300)
    return float(np.max(np.abs(a - b) / scale))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    denom = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / denom)) if a.size else 0.0
