#!/usr/bin/env python3
"""Independent second definition of the likelihood for 20 and 61 states:
fixtures that neither engine's code took part in.

  * Transition matrices P(t) = exp(Q rho t) come from mpmath at 60 significant digits
    (symmetrised eigen-decomposition mp.eigsy, cross-checked against mp.expm's Taylor
    series on the matrices with the smallest entries), rounded once to fp64: every entry --
    also the ~1e-13 three-step entries of a codon matrix -- is correct to 1 ulp.  Neither
    engine's eigen-solver, nor numpy's, is involved.
  * Discrete-Gamma rates (mean mode, Yang 1994) from scipy.special.
  * Felsenstein pruning by brute force in numpy fp64 on those matrices: sums of non-negative
    terms only, so CLVs and site likelihoods carry ~1e-15 relative error.  No scaling
    (7 taxa); with and without a proportion of invariant sites.

Outputs tests/golden/expm_fixtures.npz (inputs + expected numbers; of the 11 x 4 matrices and
5 inner CLVs per case only a spread is stored, the per-site lnL covers the rest).  Run in the build
container only (needs mpmath + scipy):   python tests/golden/make_expm_fixtures.py
Formulas: SURVEY.md Appendix B (1-7) and section 8a (a4, a5: rate/(1-pinv), the pinv term).
"""
import os
import sys

import mpmath as mp
import numpy as np
from scipy import special

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import pllhip_ctypes as pc  # noqa: E402  (input generators only: model tables, tip states)

mp.mp.dps = 60
NTIPS, NSITES, R = 7, 40, 4
KEEP_EDGES = (1, 5, 9)        # P-matrices stored: t = 0.001, 0.9, 0.0045 (all four rates)
KEEP_NODES = (7, 9, 11)       # inner CLVs stored
# an unrooted 7-taxon tree: inner nodes 7..11; (child, parent) edges with lengths that span
# 1e-3 .. 0.9 so that both near-identity and well-mixed matrices occur
EDGES = [(0, 7, 0.11), (1, 7, 0.001), (7, 8, 0.05), (2, 8, 0.4), (8, 9, 0.02), (3, 9, 0.9),
         (9, 10, 0.013), (4, 10, 0.07), (10, 11, 0.3), (5, 11, 0.0045), (6, 11, 0.21)]
# operations towards the root edge (10, 11): parent, (child, edge index) x 2
OPS = [(7, (0, 0), (1, 1)), (8, (7, 2), (2, 3)), (9, (8, 4), (3, 5)), (10, (9, 6), (4, 7)),
       (11, (5, 9), (6, 10))]
ROOT_EDGE = (10, 11, 8)      # lnL = sum_i pi_i clv10[i] sum_j P8[i][j] clv11[j]


def gamma_rates(alpha, k):
    """mean-mode discrete Gamma"""
    cuts = special.gammaincinv(alpha, np.arange(1, k) / k) / alpha
    upper = np.concatenate(([0.0], special.gammainc(alpha + 1.0, cuts * alpha), [1.0]))
    return k * np.diff(upper)


def rate_matrix(subst, freqs):
    S = len(freqs)
    Q = [[mp.mpf(0)] * S for _ in range(S)]
    it = iter(subst)
    for i in range(S):
        for j in range(i + 1, S):
            s = mp.mpf(float(next(it)))
            Q[i][j] = s * mp.mpf(float(freqs[j]))
            Q[j][i] = s * mp.mpf(float(freqs[i]))
    for i in range(S):
        Q[i][i] = -sum(Q[i][j] for j in range(S) if j != i)
    mu = -sum(mp.mpf(float(freqs[i])) * Q[i][i] for i in range(S))
    return mp.matrix([[Q[i][j] / mu for j in range(S)] for i in range(S)])


class Expm:
    """exp(Q t) at 60 digits through the symmetrised eigen-system"""

    def __init__(self, Q, freqs):
        S = Q.rows
        self.S = S
        d = [mp.sqrt(mp.mpf(float(f))) for f in freqs]
        A = mp.matrix(S, S)
        for i in range(S):
            for j in range(S):
                A[i, j] = d[i] * Q[i, j] / d[j]
        A = (A + A.T) / 2
        self.lam, U = mp.eigsy(A)
        self.V = mp.matrix(S, S)
        self.Vi = mp.matrix(S, S)
        for i in range(S):
            for k in range(S):
                self.V[i, k] = U[i, k] / d[i]
                self.Vi[k, i] = U[i, k] * d[i]

    def __call__(self, t):
        S = self.S
        e = [mp.exp(self.lam[k] * mp.mpf(t)) for k in range(S)]
        VE = mp.matrix(S, S)
        for i in range(S):
            for k in range(S):
                VE[i, k] = self.V[i, k] * e[k]
        P = VE * self.Vi
        return np.array([[float(P[i, j]) for j in range(S)] for i in range(S)])


def make_case(name, states, subst, freqs, alpha, pinv, seed):
    print(f"[{name}] rate matrix + eigen-system at {mp.mp.dps} digits ...", flush=True)
    Q = rate_matrix(subst, freqs)
    ex = Expm(Q, freqs)
    rates = gamma_rates(alpha, R)
    weights = np.full(R, 1.0 / R)
    P = np.zeros((len(EDGES), R, states, states))
    for k, (_, _, t) in enumerate(EDGES):
        for r in range(R):
            P[k, r] = ex(float(rates[r]) * t / (1.0 - pinv))
    # cross-check against an independent algorithm (Taylor series with scaling) on the two
    # matrices with the smallest entries
    for k, r in ((1, 0), (9, 0)):
        t = float(rates[r]) * EDGES[k][2] / (1.0 - pinv)
        T = mp.expm(Q * mp.mpf(t), method="taylor")
        Tn = np.array([[float(T[i, j]) for j in range(states)] for i in range(states)])
        assert np.allclose(Tn, P[k, r], rtol=1e-14, atol=1e-300), (name, k, r)
    assert P.min() >= 0.0 and np.allclose(P.sum(axis=3), 1.0, atol=1e-14)
    print(f"[{name}] smallest P entry {P[P > 0].min():.3e}", flush=True)

    rng = np.random.RandomState(seed)
    codes = rng.randint(0, states, size=(NTIPS, NSITES)).astype(np.uint8)
    codes[:, :8] = codes[0, :8]                  # a few constant columns: invariant sites
    gaps = rng.rand(NTIPS, NSITES) < 0.03
    gaps[:, :8] = False
    tipclv = np.zeros((NTIPS, NSITES, states))
    for t in range(NTIPS):
        tipclv[t, np.arange(NSITES), codes[t]] = 1.0
        tipclv[t, gaps[t]] = 1.0                  # gap = every state

    clv = {t: np.repeat(tipclv[t][:, None, :], R, axis=1) for t in range(NTIPS)}   # [site][rate][state]
    for parent, (c1, e1), (c2, e2) in OPS:
        a = np.einsum("rij,nrj->nri", P[e1], clv[c1])
        b = np.einsum("rij,nrj->nri", P[e2], clv[c2])
        clv[parent] = a * b
    pa, ch, e = ROOT_EDGE
    per_rate = np.einsum("i,nri,nri->nr", np.asarray(freqs), clv[pa], np.einsum("rij,nrj->nri", P[e], clv[ch]))
    site = (per_rate * weights).sum(axis=1)
    # invariant state: lowest state compatible with every tip, -1 if none
    common = np.ones((NSITES, states), dtype=bool)
    for t in range(NTIPS):
        common &= tipclv[t] > 0
    inv_state = np.where(common.any(axis=1), common.argmax(axis=1), -1)
    if pinv > 0:
        inv_term = np.where(inv_state >= 0, np.asarray(freqs)[np.maximum(inv_state, 0)], 0.0)
        site = (1.0 - pinv) * site + pinv * inv_term
    persite = np.log(site)
    return {
        f"{name}_states": states, f"{name}_subst": np.asarray(subst, dtype=float),
        f"{name}_freqs": np.asarray(freqs, dtype=float), f"{name}_rates": rates,
        f"{name}_weights": weights, f"{name}_pinv": pinv,
        f"{name}_codes": codes, f"{name}_gaps": gaps,
        f"{name}_pmatrix": P[list(KEEP_EDGES)], f"{name}_persite_lnl": persite, f"{name}_lnl": persite.sum(),
        f"{name}_clv_inner": np.stack([clv[n] for n in KEEP_NODES]),
        f"{name}_invariant": inv_state,
    }


def make_mixture_case(name, states, models, indices, rates, weights, seed, deriv_lengths=(0.0045, 0.3, 1.7)):
    """Mixture models: rate category r evolves under rate matrix indices[r] -- its own exchangeabilities,
    frequencies and proportion of invariant sites (libpll-2's params_indices / freqs_indices, which
    pll-modules carries per partition, src/tree/treeinfo.c:288-306, and hands to every kernel call,
    src/optimize/pll_optimize.c:79, 192, 284-346; LG4M / LG4X, src/util/models_aa.c:57-67).
    models: list of (subst, freqs, pinv); rates / weights are free (LG4X-style).
    Also stored: d(-lnL)/dt and d2(-lnL)/dt2 at the root edge for a few lengths, from P, QP, Q^2 P
    at 60 digits (what pll_update_sumtable + pll_compute_likelihood_derivatives must reproduce)."""
    print(f"[{name}] {len(models)} rate matrices at {mp.mp.dps} digits ...", flush=True)
    nm = len(models)
    Qs = [rate_matrix(sub, fr) for sub, fr, _ in models]
    exs = [Expm(Qs[m], models[m][1]) for m in range(nm)]
    freqs = [np.asarray(models[m][1], dtype=float) for m in range(nm)]
    pinvs = [float(models[m][2]) for m in range(nm)]
    Rn = len(indices)
    rho = [float(rates[r]) / (1.0 - pinvs[indices[r]]) for r in range(Rn)]
    P = np.zeros((len(EDGES), Rn, states, states))
    for k, (_, _, t) in enumerate(EDGES):
        for r in range(Rn):
            P[k, r] = exs[indices[r]](rho[r] * t)
    assert P.min() >= 0.0 and np.allclose(P.sum(axis=3), 1.0, atol=1e-14)

    rng = np.random.RandomState(seed)
    codes = rng.randint(0, states, size=(NTIPS, NSITES)).astype(np.uint8)
    codes[:, :8] = codes[0, :8]
    gaps = rng.rand(NTIPS, NSITES) < 0.03
    gaps[:, :8] = False
    tipclv = np.zeros((NTIPS, NSITES, states))
    for t in range(NTIPS):
        tipclv[t, np.arange(NSITES), codes[t]] = 1.0
        tipclv[t, gaps[t]] = 1.0
    clv = {t: np.repeat(tipclv[t][:, None, :], Rn, axis=1) for t in range(NTIPS)}
    for parent, (c1, e1), (c2, e2) in OPS:
        clv[parent] = np.einsum("rij,nrj->nri", P[e1], clv[c1]) * np.einsum("rij,nrj->nri", P[e2], clv[c2])
    pa, ch, e = ROOT_EDGE
    F = np.stack([freqs[indices[r]] for r in range(Rn)])                 # [rate][state]
    common = np.ones((NSITES, states), dtype=bool)
    for t in range(NTIPS):
        common &= tipclv[t] > 0
    inv_state = np.where(common.any(axis=1), common.argmax(axis=1), -1)
    w = np.asarray(weights, dtype=float)
    pv = np.array([pinvs[indices[r]] for r in range(Rn)])
    inv_term = np.zeros(NSITES)
    for r in range(Rn):
        inv_term += np.where(inv_state >= 0, w[r] * pv[r] * F[r][np.maximum(inv_state, 0)], 0.0)

    def site_terms(Pe):
        """[site] sum_r w_r (1 - pinv_r) sum_i pi_r[i] clv_pa[i] sum_j Pe[r][i][j] clv_ch[j]"""
        per_rate = np.einsum("ri,nri,nri->nr", F, clv[pa], np.einsum("rij,nrj->nri", Pe, clv[ch]))
        return (per_rate * w * (1.0 - pv)).sum(axis=1)

    site = site_terms(P[e]) + inv_term
    persite = np.log(site)
    # derivatives at the root edge: P(t), Q rho P(t), (Q rho)^2 P(t) per rate at 60 digits
    df, ddf = [], []
    for t in deriv_lengths:
        P0 = np.zeros((Rn, states, states)); P1 = np.zeros_like(P0); P2 = np.zeros_like(P0)
        for r in range(Rn):
            m = indices[r]
            ex = exs[m]
            el = [mp.exp(ex.lam[k] * mp.mpf(rho[r]) * mp.mpf(t)) for k in range(states)]
            for order, dst in ((0, P0), (1, P1), (2, P2)):
                VE = mp.matrix(states, states)
                for i in range(states):
                    for k in range(states):
                        VE[i, k] = ex.V[i, k] * el[k] * (ex.lam[k] * mp.mpf(rho[r])) ** order
                M = VE * ex.Vi
                dst[r] = np.array([[float(M[i, j]) for j in range(states)] for i in range(states)])
        A = site_terms(P0) + inv_term
        B = site_terms(P1)
        Cc = site_terms(P2)
        df.append(float(-(B / A).sum()))
        ddf.append(float(((B / A) ** 2 - Cc / A).sum()))
    return {
        f"{name}_states": states, f"{name}_nmodels": nm, f"{name}_indices": np.asarray(indices),
        f"{name}_subst": np.stack([np.asarray(m[0], dtype=float) for m in models]),
        f"{name}_freqs": np.stack(freqs), f"{name}_pinv": np.asarray(pinvs),
        f"{name}_rates": np.asarray(rates, dtype=float), f"{name}_weights": w,
        f"{name}_codes": codes, f"{name}_gaps": gaps,
        f"{name}_pmatrix": P[list(KEEP_EDGES)], f"{name}_persite_lnl": persite, f"{name}_lnl": persite.sum(),
        f"{name}_clv_inner": np.stack([clv[n] for n in KEEP_NODES]), f"{name}_invariant": inv_state,
        f"{name}_deriv_lengths": np.asarray(deriv_lengths), f"{name}_df": np.asarray(df), f"{name}_ddf": np.asarray(ddf),
    }


def perturbed_model(base_subst, nstates, seed):
    """a model of its own per mixture component: exchangeabilities x log-normal noise, fresh Dirichlet frequencies"""
    rng = np.random.RandomState(seed)
    sub = np.asarray(base_subst, dtype=float) * np.exp(0.5 * rng.randn(len(base_subst)))
    sub[-1] = 1.0
    fr = rng.gamma(4.0, size=nstates)
    return sub, fr / fr.sum()


def main_mixtures():
    """tests/golden/mixture_fixtures.npz: 2- and 4-matrix mixtures (20, 4 and 61 states)"""
    out = {"edges": np.array([(a, b) for a, b, _ in EDGES]), "brlens": np.array([t for _, _, t in EDGES]),
           "ops": np.array([(p, c1, e1, c2, e2) for p, (c1, e1), (c2, e2) in OPS]),
           "root_edge": np.array(ROOT_EDGE), "keep_edges": np.array(KEEP_EDGES), "keep_nodes": np.array(KEEP_NODES),
           "cases": np.array(["aa_mix2", "aa_mix4", "dna_mix2", "codon_mix2"])}
    psub, pfreq = pc.protein_model()
    csub, cfreq = pc.codon_model()
    aa = [perturbed_model(psub, 20, 100 + m) for m in range(4)]
    # two matrices alternating over four Gamma categories, each with its own p-inv
    out.update(make_mixture_case("aa_mix2", 20, [(aa[0][0], aa[0][1], 0.0), (aa[1][0], aa[1][1], 0.15)],
                                 [0, 1, 0, 1], gamma_rates(0.6, 4), np.full(4, 0.25), 11))
    # LG4X-shaped: one matrix per category, free rates and weights (weights sum to 1, mean rate 1)
    w4 = np.array([0.15, 0.35, 0.3, 0.2])
    r4 = np.array([0.2, 0.6, 1.1, 2.3])
    r4 = r4 / (r4 * w4).sum()
    out.update(make_mixture_case("aa_mix4", 20, [(aa[m][0], aa[m][1], (0.0, 0.1, 0.0, 0.05)[m]) for m in range(4)],
                                 [0, 1, 2, 3], r4, w4, 12))
    dna = [perturbed_model(pc.DNA_GTR_RATES, 4, 200 + m) for m in range(2)]
    out.update(make_mixture_case("dna_mix2", 4, [(dna[0][0], dna[0][1], 0.1), (dna[1][0], dna[1][1], 0.0)],
                                 [1, 0, 0, 1], gamma_rates(0.8, 4), np.full(4, 0.25), 13))
    c2 = pc.codon_model(kappa=3.5, omega=0.7, seed_freqs=58)
    out.update(make_mixture_case("codon_mix2", 61, [(csub, cfreq, 0.0), (c2[0], c2[1], 0.0)],
                                 [0, 1, 0, 1], gamma_rates(0.5, 4), np.full(4, 0.25), 14))
    path = os.path.join(HERE, "mixture_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    out = {"edges": np.array([(a, b) for a, b, _ in EDGES]), "brlens": np.array([t for _, _, t in EDGES]),
           "ops": np.array([(p, c1, e1, c2, e2) for p, (c1, e1), (c2, e2) in OPS]),
           "root_edge": np.array(ROOT_EDGE), "keep_edges": np.array(KEEP_EDGES), "keep_nodes": np.array(KEEP_NODES),
           "cases": np.array(["aa", "aa_pinv", "codon", "codon_pinv"])}
    psub, pfreq = pc.protein_model()
    csub, cfreq = pc.codon_model()
    out.update(make_case("aa", 20, psub, pfreq, 0.5, 0.0, 1))
    out.update(make_case("aa_pinv", 20, psub, pfreq, 0.5, 0.2, 2))
    out.update(make_case("codon", 61, csub, cfreq, 0.5, 0.0, 3))
    out.update(make_case("codon_pinv", 61, csub, cfreq, 0.5, 0.15, 4))
    path = os.path.join(HERE, "expm_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mixtures":
        main_mixtures()          # python tests/golden/make_expm_fixtures.py mixtures
    else:
        main()
