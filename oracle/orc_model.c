/*
 * oracle/orc_model.c -- substitution-model side of the CPU oracle: rate matrix,
 * eigen-decomposition, discrete gamma categories (TEST INFRASTRUCTURE ONLY).
 *
 * The reference only shows the call contracts (pll_update_prob_matrices
 * triggers the decomposition when eigen_decomp_valid == 0:
 * src/tree/treeinfo.c:854, src/algorithm/algo_callback.c:44-68;
 * pll_compute_gamma_cats(alpha, K, out, PLL_GAMMA_RATES_MEAN):
 * src/optimize/pll_optimize.c:215).  The arithmetic follows SURVEY.md
 * Appendix B steps 1-3; it is pinned by the twelve printed P-matrices and the
 * lnL of each golden file.
 *
 * Discrete gamma: the algorithm libpll-2 (xflouris/libpll-2, src/gamma.c, not
 * in the container) implements is Yang (1994) J Mol Evol 39:306, built from
 * two published routines: AS 91 (Best & Roberts 1975, chi-square quantile) and
 * AS 32 (Bhattacharjee 1970, incomplete gamma ratio).  Both are restated here
 * from the papers, with the loose termination constants of those papers
 * (that looseness is visible at ~1e-7 relative in the category rates and is
 * needed to reproduce the golden lnL to all printed digits).
 */
#include "orc_internal.h"

/* ------------------------------------------------------------------ */
/* symmetric eigenproblem: cyclic Jacobi, in extended precision         */
/* ------------------------------------------------------------------ */
/* The oracle is the checker, so it is made as exact as fp64 storage allows: the
   decomposition runs in long double (x87, 64-bit mantissa) until no rotation is left,
   and its results are rounded to fp64 once.  Measured against 60-digit matrix
   exponentials (tests/golden/make_expm_fixtures.py) the P-matrices built from it are
   correct to ~1e-16 absolute at 61 states; an fp64 Jacobi reached 1.6e-14, which showed
   as 1e-7 in single-site log-likelihoods of a codon alignment. */
typedef long double orc_ld;

void orc_jacobi_eigen(orc_ld * a, unsigned int n, orc_ld * w, orc_ld * v)
{
  unsigned int i, j, k, sweep;
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0L : 0.0L;

  for (sweep = 0; sweep < 200; ++sweep)
  {
    int rotated = 0;
    for (i = 0; i + 1 < n; ++i)
      for (j = i + 1; j < n; ++j)
      {
        orc_ld apq = a[i * n + j];
        /* a rotation that no longer changes the diagonal is skipped */
        if (apq == 0.0L ||
            (fabsl(a[i * n + i]) + fabsl(apq) * 1e-4L == fabsl(a[i * n + i]) &&
             fabsl(a[j * n + j]) + fabsl(apq) * 1e-4L == fabsl(a[j * n + j])))
        {
          a[i * n + j] = a[j * n + i] = 0.0L;
          continue;
        }
        rotated = 1;
        orc_ld theta = (a[j * n + j] - a[i * n + i]) / (2.0L * apq);
        orc_ld t = (theta >= 0 ? 1.0L : -1.0L) / (fabsl(theta) + sqrtl(theta * theta + 1.0L));
        orc_ld c = 1.0L / sqrtl(t * t + 1.0L), s = t * c;
        for (k = 0; k < n; ++k)
        {
          orc_ld akp = a[k * n + i], akq = a[k * n + j];
          a[k * n + i] = c * akp - s * akq;
          a[k * n + j] = s * akp + c * akq;
        }
        for (k = 0; k < n; ++k)
        {
          orc_ld apk = a[i * n + k], aqk = a[j * n + k];
          a[i * n + k] = c * apk - s * aqk;
          a[j * n + k] = s * apk + c * aqk;
        }
        for (k = 0; k < n; ++k)
        {
          orc_ld vkp = v[k * n + i], vkq = v[k * n + j];
          v[k * n + i] = c * vkp - s * vkq;
          v[k * n + j] = s * vkp + c * vkq;
        }
      }
    if (!rotated) break;
  }
  for (i = 0; i < n; ++i) w[i] = a[i * n + i];
}

/* Q from exchangeabilities (upper triangle, row-major) and frequencies,
   normalised to one expected substitution per unit time; then
   A = D^1/2 Q D^-1/2 = U L U^T,  V = D^-1/2 U,  V^-1 = U^T D^1/2.
   libpll-2 storage: inv_eigenvecs[i*Sp + k] = V[i][k], eigenvecs[k*Sp + j] = V^-1[k][j]
   (P = inv_eigenvecs * diag * eigenvecs). */
int orc_update_eigen(pll_partition_t * p, unsigned int idx)
{
  unsigned int S = p->states, Sp = p->states_padded, i, j, k;
  const double * pi = p->frequencies[idx];
  const double * ex = p->subst_params[idx];
  orc_ld * q = (orc_ld *)calloc((size_t)S * S, sizeof(orc_ld));
  orc_ld * u = (orc_ld *)calloc((size_t)S * S, sizeof(orc_ld));
  orc_ld * w = (orc_ld *)calloc(S, sizeof(orc_ld));
  if (!q || !u || !w)
  {
    free(q); free(u); free(w);
    orc_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate eigen workspace");
    return PLL_FAILURE;
  }

  for (i = 0, k = 0; i < S; ++i)
    for (j = i + 1; j < S; ++j, ++k)
    {
      q[i * S + j] = (orc_ld)ex[k] * pi[j];
      q[j * S + i] = (orc_ld)ex[k] * pi[i];
    }
  orc_ld mean = 0.0L;
  for (i = 0; i < S; ++i)
  {
    orc_ld row = 0.0L;
    for (j = 0; j < S; ++j) if (j != i) row += q[i * S + j];
    q[i * S + i] = -row;
    mean += pi[i] * row;
  }
  if (!(mean > 0.0L))
  {
    free(q); free(u); free(w);
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Degenerate rate matrix");
    return PLL_FAILURE;
  }
  /* symmetrise: a_ij = sqrt(pi_i) q_ij / sqrt(pi_j) / mean */
  for (i = 0; i < S; ++i)
    for (j = 0; j < S; ++j)
      q[i * S + j] = (pi[i] > 0 && pi[j] > 0)
                         ? q[i * S + j] * sqrtl((orc_ld)pi[i]) / sqrtl((orc_ld)pi[j]) / mean
                         : 0.0L;
  /* enforce exact symmetry before Jacobi */
  for (i = 0; i < S; ++i)
    for (j = i + 1; j < S; ++j)
    {
      orc_ld m = 0.5L * (q[i * S + j] + q[j * S + i]);
      q[i * S + j] = q[j * S + i] = m;
    }
  orc_jacobi_eigen(q, S, w, u);

  memset(p->eigenvecs[idx], 0, sizeof(double) * S * Sp);
  memset(p->inv_eigenvecs[idx], 0, sizeof(double) * S * Sp);
  memset(p->eigenvals[idx], 0, sizeof(double) * Sp);
  for (k = 0; k < S; ++k) p->eigenvals[idx][k] = (double)w[k];
  for (i = 0; i < S; ++i)
    for (k = 0; k < S; ++k)
    {
      orc_ld sq = sqrtl((orc_ld)pi[i]);
      p->inv_eigenvecs[idx][i * Sp + k] = (sq > 0) ? (double)(u[i * S + k] / sq) : 0.0;
      p->eigenvecs[idx][k * Sp + i] = (double)(u[i * S + k] * sq);
    }
  p->eigen_decomp_valid[idx] = 1;
  free(q); free(u); free(w);
  return PLL_SUCCESS;
}

int pll_update_eigen(pll_partition_t * p, unsigned int idx)
{
  return orc_update_eigen(p, idx);
}

/* ------------------------------------------------------------------ */
/* discrete gamma (Yang 1994)                                         */
/* ------------------------------------------------------------------ */

/* ln Gamma(x), Stirling series with recurrence shift (as used alongside
   AS 91 / AS 32 in the phylogenetics literature) */
static double orc_lngamma(double x)
{
  double f = 0.0, z;
  if (x < 7.0)
  {
    f = 1.0;
    z = x - 1.0;
    while (++z < 7.0) f *= z;
    x = z;
    f = -log(f);
  }
  z = 1.0 / (x * x);
  return f + (x - 0.5) * log(x) - x + 0.918938533204673 +
         (((-0.000595238095238 * z + 0.000793650793651) * z -
           0.002777777777778) * z + 0.083333333333333) / x;
}

/* AS 32: regularised lower incomplete gamma P(alpha, x) */
static double orc_incomplete_gamma(double x, double alpha, double lnga)
{
  const double accurate = 1e-8, overflow = 1e30;
  double factor, gin, term, rn;
  if (x == 0.0) return 0.0;
  if (x < 0.0 || alpha <= 0.0) return -1.0;

  factor = exp(alpha * log(x) - x - lnga);
  if (!(x > 1.0 && x >= alpha))
  {
    /* series */
    gin = 1.0; term = 1.0; rn = alpha;
    do
    {
      rn += 1.0;
      term *= x / rn;
      gin += term;
    } while (term > accurate);
    return gin * factor / alpha;
  }
  /* continued fraction */
  double a = 1.0 - alpha, b = a + x + 1.0, pn[6], dif;
  int i;
  term = 0.0;
  pn[0] = 1.0; pn[1] = x; pn[2] = x + 1.0; pn[3] = x * b;
  gin = pn[2] / pn[3];
  for (;;)
  {
    a += 1.0; b += 2.0; term += 1.0;
    double an = a * term;
    for (i = 0; i < 2; ++i) pn[i + 4] = b * pn[i + 2] - an * pn[i];
    if (pn[5] != 0.0)
    {
      rn = pn[4] / pn[5];
      dif = fabs(gin - rn);
      if (dif <= accurate && dif <= accurate * rn) break;
      gin = rn;
    }
    for (i = 0; i < 4; ++i) pn[i] = pn[i + 2];
    if (fabs(pn[4]) >= overflow)
      for (i = 0; i < 4; ++i) pn[i] /= overflow;
  }
  return 1.0 - factor * gin;
}

/* standard normal quantile (Odeh & Evans 1974 rational approximation, the
   starting value AS 91 asks for) */
static double orc_point_normal(double prob)
{
  const double a0 = -0.322232431088, a1 = -1.0, a2 = -0.342242088547,
               a3 = -0.0204231210245, a4 = -0.453642210148e-4;
  const double b0 = 0.0993484626060, b1 = 0.588581570495,
               b2 = 0.531103462366, b3 = 0.103537752850, b4 = 0.0038560700634;
  double p1 = (prob < 0.5) ? prob : 1.0 - prob;
  if (p1 < 1e-20) return -9999.0;
  double y = sqrt(log(1.0 / (p1 * p1)));
  double z = y + ((((y * a4 + a3) * y + a2) * y + a1) * y + a0) /
                     ((((y * b4 + b3) * y + b2) * y + b1) * y + b0);
  return (prob < 0.5) ? -z : z;
}

/* AS 91: chi-square quantile with v degrees of freedom */
static double orc_point_chi2(double prob, double v)
{
  const double e = 0.5e-6, aa = 0.6931471805, small = 1e-6;
  double p = prob, g, xx, c, ch, a, q, p1, p2, t, x, b;
  double s1, s2, s3, s4, s5, s6;
  if (p < small) return 0.0;
  if (p > 1.0 - small) return 9999.0;
  if (v <= 0.0) return -1.0;

  g = orc_lngamma(v / 2.0);
  xx = v / 2.0;
  c = xx - 1.0;
  if (v < -1.24 * log(p))
  {
    ch = pow(p * xx * exp(g + xx * aa), 1.0 / xx);
    if (ch - e < 0.0) return ch;
  }
  else if (v <= 0.32)
  {
    ch = 0.4;
    a = log(1.0 - p);
    do
    {
      q = ch;
      p1 = 1.0 + ch * (4.67 + ch);
      p2 = ch * (6.73 + ch * (6.66 + ch));
      t = -0.5 + (4.67 + 2.0 * ch) / p1 - (6.73 + ch * (13.32 + 3.0 * ch)) / p2;
      ch -= (1.0 - exp(a + g + 0.5 * ch + c * aa) * p2 / p1) / t;
    } while (fabs(q / ch - 1.0) - 0.01 > 0.0);
  }
  else
  {
    x = orc_point_normal(p);
    p1 = 0.222222 / v;
    ch = v * pow(x * sqrt(p1) + 1.0 - p1, 3.0);
    if (ch > 2.2 * v + 6.0)
      ch = -2.0 * (log(1.0 - p) - c * log(0.5 * ch) + g);
  }
  do
  {
    q = ch;
    p1 = 0.5 * ch;
    if ((t = orc_incomplete_gamma(p1, xx, g)) < 0.0) return -1.0;
    p2 = p - t;
    t = p2 * exp(xx * aa + g + p1 - c * log(ch));
    b = t / ch;
    a = 0.5 * t - b * c;
    s1 = (210.0 + a * (140.0 + a * (105.0 + a * (84.0 + a * (70.0 + 60.0 * a))))) / 420.0;
    s2 = (420.0 + a * (735.0 + a * (966.0 + a * (1141.0 + 1278.0 * a)))) / 2520.0;
    s3 = (210.0 + a * (462.0 + a * (707.0 + 932.0 * a))) / 2520.0;
    s4 = (252.0 + a * (672.0 + 1182.0 * a) + c * (294.0 + a * (889.0 + 1740.0 * a))) / 5040.0;
    s5 = (84.0 + 264.0 * a + c * (175.0 + 606.0 * a)) / 2520.0;
    s6 = (120.0 + c * (346.0 + 127.0 * c)) / 5040.0;
    ch += t * (1.0 + 0.5 * t * s1 -
               b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
  } while (fabs(q / ch - 1.0) > e);
  return ch;
}

int pll_compute_gamma_cats(double alpha, unsigned int K, double * out, int mode)
{
  unsigned int i;
  if (alpha <= 0.0 || !K || !out)
  {
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid alpha or category count");
    return PLL_FAILURE;
  }
  if (K == 1) { out[0] = 1.0; return PLL_SUCCESS; }
  double beta = alpha;

  if (mode == PLL_GAMMA_RATES_MEAN)
  {
    double lnga1 = orc_lngamma(alpha + 1.0);
    double factor = alpha / beta * K;
    double * cut = (double *)malloc(sizeof(double) * K);
    if (!cut) { orc_set_error(PLL_ERROR_MEM_ALLOC, "gamma cats"); return PLL_FAILURE; }
    for (i = 0; i + 1 < K; ++i)
      cut[i] = orc_point_chi2((i + 1.0) / K, 2.0 * alpha) / (2.0 * beta);
    for (i = 0; i + 1 < K; ++i)
      cut[i] = orc_incomplete_gamma(cut[i] * beta, alpha + 1.0, lnga1);
    out[0] = cut[0] * factor;
    out[K - 1] = (1.0 - cut[K - 2]) * factor;
    for (i = 1; i + 1 < K; ++i) out[i] = (cut[i] - cut[i - 1]) * factor;
    free(cut);
  }
  else if (mode == PLL_GAMMA_RATES_MEDIAN)
  {
    double sum = 0.0;
    for (i = 0; i < K; ++i)
    {
      out[i] = orc_point_chi2((2.0 * i + 1.0) / (2.0 * K), 2.0 * alpha) / (2.0 * beta);
      sum += out[i];
    }
    for (i = 0; i < K; ++i) out[i] *= K / sum;
  }
  else
  {
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid gamma rates mode");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}
