// pll_model.cpp -- host side of the substitution model: reversible rate matrix
// -> eigen-decomposition (consumed by the P-matrix, sumtable and derivative
// kernels) and the discrete-gamma category rates.
//
// Contracts (reference call sites):
//   eigen refresh is triggered by eigen_decomp_valid[i] == 0 on entry to
//   pll_update_prob_matrices        src/tree/treeinfo.c:854,
//                                   src/algorithm/algo_callback.c:44-68
//   pll_compute_gamma_cats(alpha, K, out, mode)
//                                   src/optimize/pll_optimize.c:215,
//                                   src/algorithm/algo_callback.c:138-141
//
// Maths: SURVEY.md Appendix B steps 1-3.  The symmetric eigenproblem is solved
// by Householder tridiagonalisation + implicit-shift QL (the oracle uses
// cyclic Jacobi, so the two engines cross-check each other).  The discrete
// gamma follows the two published routines libpll-2's gamma code is built from
// (AS 91 chi-square quantile, AS 32 incomplete gamma ratio) with the papers'
// termination constants, because pll-modules clients compare lnL values to
// ~1e-9 and those constants are visible at that level
// (test/out/optimize/blopt-minimal.out:67).
#include "engine.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace pllhip {

// ---------------------------------------------------------------------------
// symmetric eigen solver: a (n x n, row-major, symmetric) -> eigenvalues d,
// eigenvectors in the columns of z
// ---------------------------------------------------------------------------
static void tridiagonalise(std::vector<double> & a, unsigned n,
                           std::vector<double> & d, std::vector<double> & e)
{
  auto A = [&](unsigned i, unsigned j) -> double & { return a[(size_t)i * n + j]; };
  for (unsigned i = n - 1; i >= 1; --i)
  {
    unsigned l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0)
    {
      for (unsigned k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
      if (scale == 0.0)
        e[i] = A(i, l);
      else
      {
        for (unsigned k = 0; k <= l; ++k)
        {
          A(i, k) /= scale;
          h += A(i, k) * A(i, k);
        }
        double f = A(i, l);
        double g = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        A(i, l) = f - g;
        f = 0.0;
        for (unsigned j = 0; j <= l; ++j)
        {
          A(j, i) = A(i, j) / h;
          g = 0.0;
          for (unsigned k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
          for (unsigned k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
          e[j] = g / h;
          f += e[j] * A(i, j);
        }
        double hh = f / (h + h);
        for (unsigned j = 0; j <= l; ++j)
        {
          f = A(i, j);
          e[j] = g = e[j] - hh * f;
          for (unsigned k = 0; k <= j; ++k) A(j, k) -= (f * e[k] + g * A(i, k));
        }
      }
    }
    else
      e[i] = A(i, l);
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (unsigned i = 0; i < n; ++i)
  {
    if (d[i] != 0.0)
      for (unsigned j = 0; j < i; ++j)
      {
        double g = 0.0;
        for (unsigned k = 0; k < i; ++k) g += A(i, k) * A(k, j);
        for (unsigned k = 0; k < i; ++k) A(k, j) -= g * A(k, i);
      }
    d[i] = A(i, i);
    A(i, i) = 1.0;
    for (unsigned j = 0; j < i; ++j) A(j, i) = A(i, j) = 0.0;
  }
}

static bool ql_implicit(std::vector<double> & d, std::vector<double> & e, unsigned n,
                        std::vector<double> & z)
{
  auto Z = [&](unsigned i, unsigned j) -> double & { return z[(size_t)i * n + j]; };
  for (unsigned i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (unsigned l = 0; l < n; ++l)
  {
    unsigned iter = 0, m;
    do
    {
      for (m = l; m + 1 < n; ++m)
      {
        double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l)
      {
        if (iter++ == 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        long i;
        for (i = (long)m - 1; i >= (long)l; --i)
        {
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0)
          {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (unsigned k = 0; k < n; ++k)
          {
            f = Z(k, i + 1);
            Z(k, i + 1) = s * Z(k, i) + c * f;
            Z(k, i) = c * Z(k, i) - s * f;
          }
        }
        if (r == 0.0 && i >= (long)l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

// Q from exchangeabilities + frequencies, normalised to mean rate 1;
// A = D^1/2 Q D^-1/2 = U L U^T;  V = D^-1/2 U;  V^-1 = U^T D^1/2.
// evecs[i*Sp + k] = V[i][k], ievecs[k*Sp + j] = V^-1[k][j]  (P(t) = V exp(L t) V^-1).
// NOTE the partition fields: libpll-2 stores V in `inv_eigenvecs` and V^-1 in `eigenvecs`
// (its P-matrix is inv_eigenvecs * diag * eigenvecs; [libpll-2 knowledge]),
// and so does this library: update_eigen_host() and pllhip_eigen_decompose() below pass the
// fields in that order.
int eigen_decompose(unsigned S, unsigned Sp, const double * ex, const double * pi,
                    double * evecs, double * ievecs, double * evals)
{
  std::vector<double> a((size_t)S * S, 0.0), d(S), e(S);

  for (unsigned i = 0, k = 0; i < S; ++i)
    for (unsigned j = i + 1; j < S; ++j, ++k)
    {
      a[(size_t)i * S + j] = ex[k] * pi[j];
      a[(size_t)j * S + i] = ex[k] * pi[i];
    }
  double mean = 0.0;
  for (unsigned i = 0; i < S; ++i)
  {
    double row = 0.0;
    for (unsigned j = 0; j < S; ++j) if (j != i) row += a[(size_t)i * S + j];
    a[(size_t)i * S + i] = -row;
    mean += pi[i] * row;
  }
  if (!(mean > 0.0))
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Degenerate rate matrix (zero mean rate)");
    return PLL_FAILURE;
  }
  std::vector<double> sq(S);
  for (unsigned i = 0; i < S; ++i) sq[i] = std::sqrt(pi[i]);
  for (unsigned i = 0; i < S; ++i)
    for (unsigned j = 0; j < S; ++j)
      a[(size_t)i * S + j] = (sq[i] > 0 && sq[j] > 0) ? a[(size_t)i * S + j] * sq[i] / sq[j] / mean : 0.0;
  for (unsigned i = 0; i < S; ++i)
    for (unsigned j = i + 1; j < S; ++j)
    {
      double m = 0.5 * (a[(size_t)i * S + j] + a[(size_t)j * S + i]);
      a[(size_t)i * S + j] = a[(size_t)j * S + i] = m;
    }

  tridiagonalise(a, S, d, e);
  if (!ql_implicit(d, e, S, a))
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Eigen-decomposition did not converge");
    return PLL_FAILURE;
  }

  memset(evecs, 0, sizeof(double) * S * Sp);
  memset(ievecs, 0, sizeof(double) * S * Sp);
  memset(evals, 0, sizeof(double) * Sp);
  for (unsigned k = 0; k < S; ++k) evals[k] = d[k];
  for (unsigned i = 0; i < S; ++i)
    for (unsigned k = 0; k < S; ++k)
    {
      const double u = a[(size_t)i * S + k];
      evecs[(size_t)i * Sp + k] = (sq[i] > 0) ? u / sq[i] : 0.0;
      ievecs[(size_t)k * Sp + i] = u * sq[i];
    }
  return PLL_SUCCESS;
}

int update_eigen_host(pll_partition_t * p, unsigned idx)
{
  if (!eigen_decompose(p->states, p->states_padded, p->subst_params[idx], p->frequencies[idx],
                       p->inv_eigenvecs[idx], p->eigenvecs[idx], p->eigenvals[idx]))
    return PLL_FAILURE;
  p->eigen_decomp_valid[idx] = 1;
  // every path that decomposes -- the engine's own ensure_eigen and the public pll_update_eigen --
  // tells the next model check to compare the eigen-systems, also inside a burst of P-matrix requests
  if (p->engine) engine_of(p)->eigen_touched = true;
  return PLL_SUCCESS;
}

// ---------------------------------------------------------------------------
// discrete gamma
// ---------------------------------------------------------------------------
namespace {

double ln_gamma(double x)
{
  double shift = 0.0;
  if (x < 7.0)
  {
    double prod = 1.0, z = x;
    while (z < 7.0) { prod *= z; z += 1.0; }
    x = z;
    shift = -std::log(prod);
  }
  const double z2 = 1.0 / (x * x);
  const double series = (((-0.000595238095238 * z2 + 0.000793650793651) * z2 -
                          0.002777777777778) * z2 + 0.083333333333333) / x;
  return shift + (x - 0.5) * std::log(x) - x + 0.918938533204673 + series;
}

// AS 32 (Bhattacharjee 1970): P(alpha, x)
double inc_gamma_ratio(double x, double alpha, double lnga)
{
  constexpr double acc = 1e-8, big = 1e30;
  if (x == 0.0) return 0.0;
  if (x < 0.0 || alpha <= 0.0) return -1.0;
  const double factor = std::exp(alpha * std::log(x) - x - lnga);
  if (!(x > 1.0 && x >= alpha))
  {
    double sum = 1.0, term = 1.0, rn = alpha;
    do { rn += 1.0; term *= x / rn; sum += term; } while (term > acc);
    return sum * factor / alpha;
  }
  double a = 1.0 - alpha, b = a + x + 1.0, term = 0.0;
  double pn[6] = {1.0, x, x + 1.0, x * b, 0.0, 0.0};
  double gin = pn[2] / pn[3];
  for (;;)
  {
    a += 1.0; b += 2.0; term += 1.0;
    const double an = a * term;
    pn[4] = b * pn[2] - an * pn[0];
    pn[5] = b * pn[3] - an * pn[1];
    if (pn[5] != 0.0)
    {
      const double rn = pn[4] / pn[5], dif = std::fabs(gin - rn);
      if (dif <= acc && dif <= acc * rn) break;
      gin = rn;
    }
    for (int i = 0; i < 4; ++i) pn[i] = pn[i + 2];
    if (std::fabs(pn[4]) >= big)
      for (int i = 0; i < 4; ++i) pn[i] /= big;
  }
  return 1.0 - factor * gin;
}

// Odeh & Evans (1974) normal quantile
double normal_quantile(double prob)
{
  const double pp = (prob < 0.5) ? prob : 1.0 - prob;
  if (pp < 1e-20) return -9999.0;
  const double y = std::sqrt(std::log(1.0 / (pp * pp)));
  const double num = (((y * (-0.453642210148e-4) + (-0.0204231210245)) * y +
                       (-0.342242088547)) * y + (-1.0)) * y + (-0.322232431088);
  const double den = (((y * 0.0038560700634 + 0.103537752850) * y + 0.531103462366) * y +
                      0.588581570495) * y + 0.0993484626060;
  const double z = y + num / den;
  return (prob < 0.5) ? -z : z;
}

// AS 91 (Best & Roberts 1975): chi-square quantile
double chi2_quantile(double p, double v)
{
  constexpr double e = 0.5e-6, ln2 = 0.6931471805, small = 1e-6;
  if (p < small) return 0.0;
  if (p > 1.0 - small) return 9999.0;
  if (v <= 0.0) return -1.0;
  const double g = ln_gamma(v / 2.0), xx = v / 2.0, c = xx - 1.0;
  double ch;
  if (v < -1.24 * std::log(p))
  {
    ch = std::pow(p * xx * std::exp(g + xx * ln2), 1.0 / xx);
    if (ch - e < 0.0) return ch;
  }
  else if (v <= 0.32)
  {
    ch = 0.4;
    const double a = std::log(1.0 - p);
    double q;
    do
    {
      q = ch;
      const double p1 = 1.0 + ch * (4.67 + ch);
      const double p2 = ch * (6.73 + ch * (6.66 + ch));
      const double t = -0.5 + (4.67 + 2.0 * ch) / p1 - (6.73 + ch * (13.32 + 3.0 * ch)) / p2;
      ch -= (1.0 - std::exp(a + g + 0.5 * ch + c * ln2) * p2 / p1) / t;
    } while (std::fabs(q / ch - 1.0) - 0.01 > 0.0);
  }
  else
  {
    const double x = normal_quantile(p), p1 = 0.222222 / v;
    ch = v * std::pow(x * std::sqrt(p1) + 1.0 - p1, 3.0);
    if (ch > 2.2 * v + 6.0) ch = -2.0 * (std::log(1.0 - p) - c * std::log(0.5 * ch) + g);
  }
  double q;
  do
  {
    q = ch;
    const double p1 = 0.5 * ch;
    double t = inc_gamma_ratio(p1, xx, g);
    if (t < 0.0) return -1.0;
    const double p2 = p - t;
    t = p2 * std::exp(xx * ln2 + g + p1 - c * std::log(ch));
    const double b = t / ch, a = 0.5 * t - b * c;
    const double s1 = (210.0 + a * (140.0 + a * (105.0 + a * (84.0 + a * (70.0 + 60.0 * a))))) / 420.0;
    const double s2 = (420.0 + a * (735.0 + a * (966.0 + a * (1141.0 + 1278.0 * a)))) / 2520.0;
    const double s3 = (210.0 + a * (462.0 + a * (707.0 + 932.0 * a))) / 2520.0;
    const double s4 = (252.0 + a * (672.0 + 1182.0 * a) + c * (294.0 + a * (889.0 + 1740.0 * a))) / 5040.0;
    const double s5 = (84.0 + 264.0 * a + c * (175.0 + 606.0 * a)) / 2520.0;
    const double s6 = (120.0 + c * (346.0 + 127.0 * c)) / 5040.0;
    ch += t * (1.0 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
  } while (std::fabs(q / ch - 1.0) > e);
  return ch;
}

} // namespace
} // namespace pllhip

extern "C" int pllhip_eigen_decompose(unsigned int states, unsigned int states_padded,
                                     const double * subst_params, const double * frequencies,
                                     double * eigenvecs, double * inv_eigenvecs, double * eigenvals)
{
  return pllhip::eigen_decompose(states, states_padded, subst_params, frequencies,
                                 inv_eigenvecs, eigenvecs, eigenvals);
}

extern "C" int pll_compute_gamma_cats(double alpha, unsigned int K, double * out, int mode)
{
  using namespace pllhip;
  if (!(alpha > 0.0) || !K || !out)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Invalid alpha or category count");
    return PLL_FAILURE;
  }
  if (K == 1) { out[0] = 1.0; return PLL_SUCCESS; }
  const double beta = alpha;
  if (mode == PLL_GAMMA_RATES_MEAN)
  {
    const double lnga1 = ln_gamma(alpha + 1.0), factor = alpha / beta * K;
    std::vector<double> cut(K);
    for (unsigned i = 0; i + 1 < K; ++i)
      cut[i] = chi2_quantile((i + 1.0) / K, 2.0 * alpha) / (2.0 * beta);
    for (unsigned i = 0; i + 1 < K; ++i)
      cut[i] = inc_gamma_ratio(cut[i] * beta, alpha + 1.0, lnga1);
    out[0] = cut[0] * factor;
    out[K - 1] = (1.0 - cut[K - 2]) * factor;
    for (unsigned i = 1; i + 1 < K; ++i) out[i] = (cut[i] - cut[i - 1]) * factor;
  }
  else if (mode == PLL_GAMMA_RATES_MEDIAN)
  {
    double sum = 0.0;
    for (unsigned i = 0; i < K; ++i)
    {
      out[i] = chi2_quantile((2.0 * i + 1.0) / (2.0 * K), 2.0 * alpha) / (2.0 * beta);
      sum += out[i];
    }
    for (unsigned i = 0; i < K; ++i) out[i] *= K / sum;
  }
  else
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Invalid gamma rates mode");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}
