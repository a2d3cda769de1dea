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
    main()
