/*
 * oracle/orc_kernels.c -- the likelihood hot path of the CPU oracle
 * (TEST INFRASTRUCTURE ONLY, see orc_internal.h).
 *
 * Each function is the scalar fp64 restatement of one row of SURVEY.md
 * section 8a; the call contract is taken from the reference call sites:
 *   pll_update_prob_matrices           src/tree/treeinfo.c:854
 *   pll_update_partials                src/tree/treeinfo.c:1037,
 *                                      src/optimize/pll_optimize.c:748-775
 *   pll_compute_edge_loglikelihood     src/tree/treeinfo.c:1049
 *   pll_compute_root_loglikelihood     src/optimize/pll_optimize.c:329
 *   pll_update_sumtable                src/optimize/pll_optimize.c:800, 1468
 *   pll_compute_likelihood_derivatives src/optimize/pll_optimize.c:307, 1249
 *                                      (sign: src/optimize/opt_algorithms.c:208-226)
 *   pll_compute_node_ancestral         src/tree/treeinfo.c:1698
 *
 * Semantics that the reference cannot pin offline and that therefore DEFINE
 * the behaviour both engines must share (libpll-2 knowledge):
 *   - scaling: if the parent has a scaler, scaler[n] = s_child1[n]+s_child2[n];
 *     if all R*S entries of site n are < 2^-256 they are multiplied by 2^256
 *     and scaler[n] += 1.  lnL subtracts 256*ln2 per count.
 *   - p-inv: P-matrices use rate/(1-pinv); site likelihood is
 *     (1-pinv)*L + pinv*pi[invariant[n]].
 *   - t == 0 gives the identity matrix.
 *   - PLL_ATTRIB_RATE_SCALERS: scale buffers hold one count per (site, rate),
 *     scaler[n*R + r]; the all-small test and the 2^256 step act on the S entries of one
 *     rate; a site's terms are brought to the smallest count among its rates
 *     before they are added: a rate that is d counts above it is multiplied by 2^(-256
 *     min(d, 4)) (libpll-2's PLL_SCALE_RATE_MAXDIFF), and the site carries that smallest count.
 */
#include "orc_internal.h"

#define ORC_LN_SCALE 177.445678223345993274 /* 256 * ln 2 */

int pll_update_prob_matrices(pll_partition_t * p,
                             const unsigned int * params_indices,
                             const unsigned int * matrix_indices,
                             const double * branch_lengths,
                             unsigned int count)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int m, r, i, j, k;
  double * ex = (double *)malloc(sizeof(double) * S);
  if (!ex) { orc_set_error(PLL_ERROR_MEM_ALLOC, "pmatrix workspace"); return PLL_FAILURE; }

  for (r = 0; r < R; ++r)
    if (!p->eigen_decomp_valid[params_indices[r]])
      if (!orc_update_eigen(p, params_indices[r])) { free(ex); return PLL_FAILURE; }

  for (m = 0; m < count; ++m)
  {
    double t = branch_lengths[m];
    if (matrix_indices[m] >= p->prob_matrices || !(t >= 0.0))
    {
      free(ex);
      orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid matrix index or branch length");
      return PLL_FAILURE;
    }
    for (r = 0; r < R; ++r)
    {
      unsigned int pi_ = params_indices[r];
      double * P = p->pmatrix[matrix_indices[m]] + (size_t)r * S * Sp;
      const double * V = p->inv_eigenvecs[pi_], * Vi = p->eigenvecs[pi_];   /* libpll-2 naming */
      const double * L = p->eigenvals[pi_];
      double pinv = p->prop_invar[pi_];
      memset(P, 0, sizeof(double) * S * Sp);
      if (t == 0.0)
      {
        for (i = 0; i < S; ++i) P[i * Sp + i] = 1.0;
        continue;
      }
      double rt = p->rates[r] * t / (1.0 - pinv);
      for (k = 0; k < S; ++k) ex[k] = exp(L[k] * rt);
      for (i = 0; i < S; ++i)
        for (j = 0; j < S; ++j)
        {
          /* extended-precision accumulation: the sum cancels down to ~1e-13 at 61 states */
          long double s = 0.0L;
          for (k = 0; k < S; ++k) s += (long double)V[i * Sp + k] * ex[k] * Vi[k * Sp + j];
          P[i * Sp + j] = (s > 0.0L) ? (double)s : 0.0;      /* a probability: cancellation noise is not allowed below 0 */
        }
    }
  }
  free(ex);
  return PLL_SUCCESS;
}

/* per-code lookup for a tip child stored as codes:
   lut[(r*ncodes + code)*S + i] = sum_j P[r][i][j] * mask(code)[j] */
static double * tip_lookup(const pll_partition_t * p, const double * P)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int nc = p->maxstates, r, c, i, j;
  double * lut = (double *)malloc(sizeof(double) * (size_t)R * (nc ? nc : 1) * S);
  if (!lut) return NULL;
  for (r = 0; r < R; ++r)
    for (c = 0; c < nc; ++c)
    {
      pll_state_t m = p->tipmap[c];
      for (i = 0; i < S; ++i)
      {
        double a = 0.0;
        for (j = 0; j < S; ++j)
          if ((m >> j) & 1ULL) a += P[((size_t)r * S + i) * Sp + j];
        lut[((size_t)r * nc + c) * S + i] = a;
      }
    }
  return lut;
}

/* ---------------------------------------------------------------------------
 * Vectorised variant for 4, 20 and 61 states (GCC vector extensions -> AVX2 + FMA with
 * -march=x86-64-v3), used when ORC_FAST=1: the P-matrix is transposed once per
 * operation and rate so that a parent vector is built as sum_j Pt[j][:] * c[j]
 * (broadcast-multiply-accumulate over whole state vectors instead of one dot product
 * per state).  Same arithmetic, different summation order: results agree with the
 * scalar path to rounding (tests/test_oracle_properties.py).  It exists so that the CPU
 * baseline bench.py reports next to the GPU numbers is not an artificially slow one;
 * the scalar path stays the reference for the golden files.
 * ------------------------------------------------------------------------- */
typedef double orc_v4 __attribute__((vector_size(32), aligned(8)));

static int orc_fast(void)
{
  static int on = -1;
  if (on < 0) { const char * e = getenv("ORC_FAST"); on = (e && atoi(e)) ? 1 : 0; }
  return on;
}

/* a[0..4NV) = sum_{j<S} Pt[j][0..4NV) * c[j];  4 NV = S rounded up to a multiple of 4 */
#define ORC_MATVEC(NV, Pt, c, a)                                        \
  do {                                                                  \
    unsigned int j_, v_;                                                \
    for (v_ = 0; v_ < NV; ++v_) a[v_] = (orc_v4){0.0, 0.0, 0.0, 0.0};   \
    for (j_ = 0; j_ < S; ++j_)                                          \
    {                                                                   \
      const double cj_ = (c)[j_];                                       \
      const orc_v4 b_ = {cj_, cj_, cj_, cj_};                           \
      const orc_v4 * row_ = (const orc_v4 *)((Pt) + j_ * 4 * NV);       \
      for (v_ = 0; v_ < NV; ++v_) a[v_] += row_[v_] * b_;               \
    }                                                                   \
  } while (0)

#define ORC_FAST_BODY(NV)                                                                     \
  for (n = 0; n < (long)orc_salloc(p); ++n)                                                   \
  {                                                                                           \
    double * out = parent + (size_t)n * R * Sp;                                               \
    int all_small = 1;                                                                        \
    unsigned int r, v;                                                                        \
    for (r = 0; r < R; ++r)                                                                   \
    {                                                                                         \
      orc_v4 a[NV], b[NV];                                                                    \
      if (tip1) { memset(a, 0, sizeof(a)); memcpy(a, lut1 + ((size_t)r * nc + code1[n]) * S, S * sizeof(double)); } \
      else ORC_MATVEC(NV, Pt1 + (size_t)r * S * 4 * NV, c1 + ((size_t)n * R + r) * Sp, a);    \
      if (tip2) { memset(b, 0, sizeof(b)); memcpy(b, lut2 + ((size_t)r * nc + code2[n]) * S, S * sizeof(double)); } \
      else ORC_MATVEC(NV, Pt2 + (size_t)r * S * 4 * NV, c2 + ((size_t)n * R + r) * Sp, b);    \
      for (v = 0; v < NV; ++v)                                                                \
      {                                                                                       \
        const orc_v4 x = a[v] * b[v];                                                         \
        const unsigned int left = Sp - 4 * v;                                                 \
        memcpy(out + r * Sp + 4 * v, &x, (left < 4 ? left : 4) * sizeof(double));             \
        if (!(x[0] < PLL_SCALE_THRESHOLD && x[1] < PLL_SCALE_THRESHOLD &&                     \
              x[2] < PLL_SCALE_THRESHOLD && x[3] < PLL_SCALE_THRESHOLD)) all_small = 0;       \
      }                                                                                       \
    }                                                                                         \
    if (ps)                                                                                   \
    {                                                                                         \
      unsigned int cnt = (s1 ? s1[n] : 0) + (s2 ? s2[n] : 0);                                 \
      if (all_small)                                                                          \
      {                                                                                       \
        unsigned int i;                                                                       \
        for (i = 0; i < R * Sp; ++i) out[i] *= PLL_SCALE_FACTOR;                              \
        cnt += 1;                                                                             \
      }                                                                                       \
      ps[n] = cnt;                                                                            \
    }                                                                                         \
  }

static void partials_fast(pll_partition_t * p, double * parent, unsigned int * ps,
                          const unsigned int * s1, const unsigned int * s2,
                          const double * P1, const double * P2, int tip1, int tip2,
                          const unsigned char * code1, const unsigned char * code2,
                          const double * c1, const double * c2,
                          const double * lut1, const double * lut2)
{
  const unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats, nc = p->maxstates;
  /* transposed, rows zero-padded to W = S rounded up to a multiple of 4: Pt[r][j][0..W) */
  const unsigned int W = (S + 3u) & ~3u;
  double * Pt1 = (double *)calloc((size_t)2 * R * S * W, sizeof(double)), * Pt2 = Pt1 + (size_t)R * S * W;
  unsigned int r, i, j;
  long n;
  for (r = 0; r < R; ++r)
    for (i = 0; i < S; ++i)
      for (j = 0; j < S; ++j)
      {
        Pt1[((size_t)r * S + j) * W + i] = P1[((size_t)r * S + i) * Sp + j];
        Pt2[((size_t)r * S + j) * W + i] = P2[((size_t)r * S + i) * Sp + j];
      }
  if (W == 64)
  {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((size_t)p->sites * R * S * S > 4000000)
#endif
    ORC_FAST_BODY(16)
  }
  else if (S == 20)
  {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((size_t)p->sites * R * S * S > 4000000)
#endif
    ORC_FAST_BODY(5)
  }
  else
  {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((size_t)p->sites * R * S * S > 4000000)
#endif
    ORC_FAST_BODY(1)
  }
  free(Pt1);
}

void pll_update_partials(pll_partition_t * p, const pll_operation_t * ops,
                         unsigned int count)
{
  const unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  const unsigned int nc = p->maxstates;
  const int coded = (p->attributes & PLL_ATTRIB_PATTERN_TIP) != 0;
  unsigned int o;

  for (o = 0; o < count; ++o)
  {
    const pll_operation_t * op = &ops[o];
    double * parent = p->clv[op->parent_clv_index];
    unsigned int * ps = (op->parent_scaler_index == PLL_SCALE_BUFFER_NONE)
                            ? NULL : p->scale_buffer[op->parent_scaler_index];
    const unsigned int * s1 = (op->child1_scaler_index == PLL_SCALE_BUFFER_NONE)
                            ? NULL : p->scale_buffer[op->child1_scaler_index];
    const unsigned int * s2 = (op->child2_scaler_index == PLL_SCALE_BUFFER_NONE)
                            ? NULL : p->scale_buffer[op->child2_scaler_index];
    const double * P1 = p->pmatrix[op->child1_matrix_index];
    const double * P2 = p->pmatrix[op->child2_matrix_index];
    const int tip1 = coded && op->child1_clv_index < p->tips;
    const int tip2 = coded && op->child2_clv_index < p->tips;
    const unsigned char * code1 = tip1 ? p->tipchars[op->child1_clv_index] : NULL;
    const unsigned char * code2 = tip2 ? p->tipchars[op->child2_clv_index] : NULL;
    const double * c1 = tip1 ? NULL : p->clv[op->child1_clv_index];
    const double * c2 = tip2 ? NULL : p->clv[op->child2_clv_index];
    double * lut1 = tip1 ? tip_lookup(p, P1) : NULL;
    double * lut2 = tip2 ? tip_lookup(p, P2) : NULL;
    if ((tip1 && !lut1) || (tip2 && !lut2))
    {
      free(lut1); free(lut2);
      orc_set_error(PLL_ERROR_MEM_ALLOC, "tip lookup");
      return;
    }

    if (orc_fast() && (S == 20 || S == 4 || S == 61) && !(p->attributes & PLL_ATTRIB_RATE_SCALERS))
    {
      partials_fast(p, parent, ps, s1, s2, P1, P2, tip1, tip2, code1, code2, c1, c2, lut1, lut2);
      free(lut1);
      free(lut2);
      continue;
    }

    long n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((size_t)p->sites * R * S * S > 4000000)
#endif
    for (n = 0; n < (long)orc_salloc(p); ++n)
    {
      double * out = parent + (size_t)n * R * Sp;
      int all_small = 1;
      unsigned int r, i, j;
      for (r = 0; r < R; ++r)
      {
        const double * v1 = tip1 ? lut1 + ((size_t)r * nc + code1[n]) * S
                                 : c1 + ((size_t)n * R + r) * Sp;
        const double * v2 = tip2 ? lut2 + ((size_t)r * nc + code2[n]) * S
                                 : c2 + ((size_t)n * R + r) * Sp;
        for (i = 0; i < S; ++i)
        {
          double a, b;
          if (tip1) a = v1[i];
          else
          {
            const double * row = P1 + ((size_t)r * S + i) * Sp;
            for (a = 0.0, j = 0; j < S; ++j) a += row[j] * v1[j];
          }
          if (tip2) b = v2[i];
          else
          {
            const double * row = P2 + ((size_t)r * S + i) * Sp;
            for (b = 0.0, j = 0; j < S; ++j) b += row[j] * v2[j];
          }
          double v = a * b;
          out[r * Sp + i] = v;
          if (!(v < PLL_SCALE_THRESHOLD)) all_small = 0;
        }
        for (i = S; i < Sp; ++i) out[r * Sp + i] = 0.0;
      }
      if (ps && (p->attributes & PLL_ATTRIB_RATE_SCALERS))
      {
        for (r = 0; r < R; ++r)
        {
          unsigned int cnt = (s1 ? s1[(size_t)n * R + r] : 0) + (s2 ? s2[(size_t)n * R + r] : 0);
          int small = 1;
          for (i = 0; i < S; ++i) if (!(out[r * Sp + i] < PLL_SCALE_THRESHOLD)) small = 0;
          if (small)
          {
            for (i = 0; i < S; ++i) out[r * Sp + i] *= PLL_SCALE_FACTOR;
            cnt += 1;
          }
          ps[(size_t)n * R + r] = cnt;
        }
      }
      else if (ps)
      {
        unsigned int cnt = (s1 ? s1[n] : 0) + (s2 ? s2[n] : 0);
        if (all_small)
        {
          for (r = 0; r < R; ++r)
            for (i = 0; i < S; ++i) out[r * Sp + i] *= PLL_SCALE_FACTOR;
          cnt += 1;
        }
        ps[n] = cnt;
      }
    }
    free(lut1);
    free(lut2);
  }
}

#define ORC_SCALE_RATE_MAXDIFF 4

/* per-rate scalers: counts of the R rates of site n (both ends of the edge), their
   minimum, and the factor that brings rate r down to it */
static unsigned int rate_counts(const pll_partition_t * p, const unsigned int * s1, const unsigned int * s2,
                                unsigned int n, double * factor)
{
  unsigned int R = p->rate_cats, r, mn = ~0u, cnt[64];
  for (r = 0; r < R; ++r)
  {
    cnt[r] = (s1 ? s1[(size_t)n * R + r] : 0) + (s2 ? s2[(size_t)n * R + r] : 0);
    if (cnt[r] < mn) mn = cnt[r];
  }
  for (r = 0; r < R; ++r)
  {
    unsigned int d = cnt[r] - mn;
    if (d > ORC_SCALE_RATE_MAXDIFF) d = ORC_SCALE_RATE_MAXDIFF;
    factor[r] = d ? ldexp(1.0, -256 * (int)d) : 1.0;
  }
  return mn;
}

/* combine the scaled site likelihood `x` (true value x * 2^(-256 cnt)), the
   invariant-site term and the pattern weight into a weighted log-likelihood */
static double site_loglh(double x, unsigned int cnt, double inv_term)
{
  if (inv_term > 0.0 && cnt > 0)
  {
    /* bring x to true scale; beyond 3 counts it underflows to 0 next to inv */
    double xt = (cnt <= 3) ? ldexp(x, -256 * (int)cnt) : 0.0;
    return log(xt + inv_term);
  }
  return log(x + inv_term) - (double)cnt * ORC_LN_SCALE;
}

/* invariant state of pattern n: the alignment's (pll_update_invariant_sites), or the state an
   ascertainment-bias column consists of */
static int inv_state(const pll_partition_t * p, unsigned int n)
{
  if (n >= p->sites) return (int)(n - p->sites);
  return p->invariant ? p->invariant[n] : -1;
}

/* ascertainment-bias correction from the log-likelihoods l[k] of the S constant patterns
   (Leache et al. 2015):
     Lewis        - W log(1 - sum_k L_k)      W = weight sum of the alignment patterns
     Felsenstein  + w log(sum_k L_k)          w = sum of the state weights
     Stamatakis   + sum_k w_k log(L_k) */
static double asc_correction(const pll_partition_t * p, const double * l)
{
  const unsigned int S = p->states, * w = p->pattern_weights + p->sites;
  unsigned int k, wsum = 0;
  double mx = l[0], sum = 0.0, corr = 0.0;
  for (k = 0; k < S; ++k) { if (l[k] > mx) mx = l[k]; wsum += w[k]; }
  for (k = 0; k < S; ++k) sum += exp(l[k] - mx);            /* sum_k L_k = exp(mx) * sum */
  switch (p->attributes & PLL_ATTRIB_AB_MASK)
  {
    case PLL_ATTRIB_AB_LEWIS:
      return -(double)p->pattern_weight_sum * log1p(-exp(mx) * sum);
    case PLL_ATTRIB_AB_FELSENSTEIN:
      return (double)wsum * (mx + log(sum));
    case PLL_ATTRIB_AB_STAMATAKIS:
      for (k = 0; k < S; ++k) corr += (double)w[k] * l[k];
      return corr;
    default:
      return 0.0;
  }
}

double pll_compute_edge_loglikelihood(pll_partition_t * p,
                                      unsigned int pc, int psc,
                                      unsigned int cc, int csc,
                                      unsigned int matrix_index,
                                      const unsigned int * freqs_indices,
                                      double * persite_lnl)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int n, r, i, j;
  const unsigned int * s1 = (psc == PLL_SCALE_BUFFER_NONE) ? NULL : p->scale_buffer[psc];
  const unsigned int * s2 = (csc == PLL_SCALE_BUFFER_NONE) ? NULL : p->scale_buffer[csc];
  const double * P = p->pmatrix[matrix_index];
  const int rate_scalers = (p->attributes & PLL_ATTRIB_RATE_SCALERS) != 0;
  double total = 0.0, asc_l[64];

  for (n = 0; n < orc_salloc(p); ++n)
  {
    double site = 0.0, inv_term = 0.0, factor[64];
    unsigned int min_cnt = rate_scalers ? rate_counts(p, s1, s2, n, factor) : 0;
    for (r = 0; r < R; ++r)
    {
      const double * pi = p->frequencies[freqs_indices[r]];
      double pinv = p->prop_invar[freqs_indices[r]];
      double lr = 0.0;
      for (i = 0; i < S; ++i)
      {
        double a = 0.0;
        for (j = 0; j < S; ++j)
          a += P[((size_t)r * S + i) * Sp + j] * orc_clv_at(p, cc, n, r, j);
        lr += pi[i] * orc_clv_at(p, pc, n, r, i) * a;
      }
      if (rate_scalers) lr *= factor[r];
      if (pinv > 0.0)
      {
        site += p->rate_weights[r] * (1.0 - pinv) * lr;
        if (inv_state(p, n) >= 0)
          inv_term += p->rate_weights[r] * pinv * pi[inv_state(p, n)];
      }
      else
        site += p->rate_weights[r] * lr;
    }
    unsigned int cnt = rate_scalers ? min_cnt : (s1 ? s1[n] : 0) + (s2 ? s2[n] : 0);
    double l = site_loglh(site, cnt, inv_term);
    if (n >= p->sites) { asc_l[n - p->sites] = l; continue; }     /* a constant pattern of the correction */
    if (persite_lnl) persite_lnl[n] = l;
    total += l * p->pattern_weights[n];
  }
  if (p->asc_bias_alloc) total += asc_correction(p, asc_l);
  return total;
}

double pll_compute_root_loglikelihood(pll_partition_t * p,
                                      unsigned int clv, int sc,
                                      const unsigned int * freqs_indices,
                                      double * persite_lnl)
{
  unsigned int S = p->states, R = p->rate_cats, n, r, i;
  const unsigned int * s1 = (sc == PLL_SCALE_BUFFER_NONE) ? NULL : p->scale_buffer[sc];
  const int rate_scalers = (p->attributes & PLL_ATTRIB_RATE_SCALERS) != 0;
  double total = 0.0, asc_l[64];
  for (n = 0; n < orc_salloc(p); ++n)
  {
    double site = 0.0, inv_term = 0.0, factor[64];
    unsigned int min_cnt = rate_scalers ? rate_counts(p, s1, NULL, n, factor) : 0;
    for (r = 0; r < R; ++r)
    {
      const double * pi = p->frequencies[freqs_indices[r]];
      double pinv = p->prop_invar[freqs_indices[r]];
      double lr = 0.0;
      for (i = 0; i < S; ++i) lr += pi[i] * orc_clv_at(p, clv, n, r, i);
      if (rate_scalers) lr *= factor[r];
      if (pinv > 0.0)
      {
        site += p->rate_weights[r] * (1.0 - pinv) * lr;
        if (inv_state(p, n) >= 0)
          inv_term += p->rate_weights[r] * pinv * pi[inv_state(p, n)];
      }
      else
        site += p->rate_weights[r] * lr;
    }
    double l = site_loglh(site, rate_scalers ? min_cnt : (s1 ? s1[n] : 0), inv_term);
    if (n >= p->sites) { asc_l[n - p->sites] = l; continue; }
    if (persite_lnl) persite_lnl[n] = l;
    total += l * p->pattern_weights[n];
  }
  if (p->asc_bias_alloc) total += asc_correction(p, asc_l);
  return total;
}

int pll_update_sumtable(pll_partition_t * p,
                        unsigned int pc, unsigned int cc,
                        int psc, int csc,
                        const unsigned int * params_indices,
                        double * sumtable)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int n, r, i, k;
  (void)psc; (void)csc;
  for (r = 0; r < R; ++r)
    if (!p->eigen_decomp_valid[params_indices[r]])
      if (!orc_update_eigen(p, params_indices[r])) return PLL_FAILURE;

  for (n = 0; n < orc_salloc(p); ++n)
    for (r = 0; r < R; ++r)
    {
      unsigned int pi_ = params_indices[r];
      const double * pi = p->frequencies[pi_];
      const double * V = p->inv_eigenvecs[pi_], * Vi = p->eigenvecs[pi_];   /* libpll-2 naming */
      double * out = sumtable + ((size_t)n * R + r) * Sp;
      for (k = 0; k < S; ++k)
      {
        double a = 0.0, b = 0.0;
        for (i = 0; i < S; ++i)
        {
          a += orc_clv_at(p, pc, n, r, i) * pi[i] * V[i * Sp + k];
          b += Vi[k * Sp + i] * orc_clv_at(p, cc, n, r, i);
        }
        out[k] = a * b;
      }
      for (k = S; k < Sp; ++k) out[k] = 0.0;
    }
  return PLL_SUCCESS;
}

/* first and second derivative (by the branch length) of the ascertainment-bias correction from
   the likelihoods A, their derivatives B, C of the S constant patterns, each carrying cnt[k]
   scaling steps:  Lewis  W [B / (1 - A)],  W [C / (1 - A) + (B / (1 - A))^2]  with A, B, C summed
   over k in true scale;  Felsenstein  w [B / A],  w [C / A - (B / A)^2]  (sums; the common scale
   cancels);  Stamatakis  sum_k w_k [B_k / A_k],  sum_k w_k [C_k / A_k - (B_k / A_k)^2] */
static void asc_derivatives(const pll_partition_t * p, const double * A, const double * B, const double * C,
                            const unsigned int * cnt, double * d1, double * d2)
{
  const unsigned int S = p->states, * w = p->pattern_weights + p->sites;
  unsigned int k, mn = ~0u, wsum = 0;
  double a = 0.0, b = 0.0, c = 0.0;
  *d1 = *d2 = 0.0;
  for (k = 0; k < S; ++k) { if (cnt[k] < mn) mn = cnt[k]; wsum += w[k]; }
  for (k = 0; k < S; ++k)
  {
    const unsigned int d = cnt[k] - mn;
    const double f = (d == 0) ? 1.0 : (d <= 3) ? ldexp(1.0, -256 * (int)d) : 0.0;
    a += f * A[k]; b += f * B[k]; c += f * C[k];
  }
  switch (p->attributes & PLL_ATTRIB_AB_MASK)
  {
    case PLL_ATTRIB_AB_LEWIS:
    {
      const double t = (mn == 0) ? 1.0 : (mn <= 3) ? ldexp(1.0, -256 * (int)mn) : 0.0;   /* to true scale */
      const double q = t * b / (1.0 - t * a);
      *d1 = (double)p->pattern_weight_sum * q;
      *d2 = (double)p->pattern_weight_sum * (t * c / (1.0 - t * a) + q * q);
      break;
    }
    case PLL_ATTRIB_AB_FELSENSTEIN:
      *d1 = (double)wsum * (b / a);
      *d2 = (double)wsum * (c / a - (b / a) * (b / a));
      break;
    case PLL_ATTRIB_AB_STAMATAKIS:
      for (k = 0; k < S; ++k)
      {
        *d1 += (double)w[k] * (B[k] / A[k]);
        *d2 += (double)w[k] * (C[k] / A[k] - (B[k] / A[k]) * (B[k] / A[k]));
      }
      break;
    default:
      break;
  }
}

int pll_compute_likelihood_derivatives(pll_partition_t * p,
                                       int psc, int csc,
                                       double t,
                                       const unsigned int * params_indices,
                                       const double * sumtable,
                                       double * d_f, double * dd_f)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int n, r, k;
  const unsigned int * s1 = (psc == PLL_SCALE_BUFFER_NONE) ? NULL : p->scale_buffer[psc];
  const unsigned int * s2 = (csc == PLL_SCALE_BUFFER_NONE) ? NULL : p->scale_buffer[csc];
  double * e0 = (double *)malloc(sizeof(double) * 3 * R * S);
  if (!e0) { orc_set_error(PLL_ERROR_MEM_ALLOC, "derivative workspace"); return PLL_FAILURE; }
  double * e1 = e0 + (size_t)R * S, * e2 = e1 + (size_t)R * S;

  for (r = 0; r < R; ++r)
  {
    unsigned int pi_ = params_indices[r];
    double pinv = p->prop_invar[pi_];
    double rho = p->rates[r] / (1.0 - pinv);
    double wr = p->rate_weights[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
    for (k = 0; k < S; ++k)
    {
      double lam = p->eigenvals[pi_][k] * rho;
      double ex = exp(lam * t);
      e0[r * S + k] = wr * ex;
      e1[r * S + k] = wr * ex * lam;
      e2[r * S + k] = wr * ex * lam * lam;
    }
  }

  const int rate_scalers = (p->attributes & PLL_ATTRIB_RATE_SCALERS) != 0;
  double df = 0.0, ddf = 0.0, ascA[64], ascB[64], ascC[64];
  unsigned int asc_cnt[64];
  for (n = 0; n < orc_salloc(p); ++n)
  {
    double A = 0.0, B = 0.0, C = 0.0, inv_term = 0.0, factor[64];
    unsigned int min_cnt = rate_scalers ? rate_counts(p, s1, s2, n, factor) : 0;
    for (r = 0; r < R; ++r)
    {
      const double * st = sumtable + ((size_t)n * R + r) * Sp;
      if (rate_scalers)
      {
        /* the rate's sums first, then its factor */
        double a = 0.0, b = 0.0, c = 0.0;
        for (k = 0; k < S; ++k)
        {
          a += st[k] * e0[r * S + k];
          b += st[k] * e1[r * S + k];
          c += st[k] * e2[r * S + k];
        }
        A += factor[r] * a;
        B += factor[r] * b;
        C += factor[r] * c;
      }
      else
      for (k = 0; k < S; ++k)
      {
        A += st[k] * e0[r * S + k];
        B += st[k] * e1[r * S + k];
        C += st[k] * e2[r * S + k];
      }
      unsigned int pi_ = params_indices[r];
      double pinv = p->prop_invar[pi_];
      if (pinv > 0.0 && inv_state(p, n) >= 0)
        inv_term += p->rate_weights[r] * pinv * p->frequencies[pi_][inv_state(p, n)];
    }
    const unsigned int cnt = rate_scalers ? min_cnt : (s1 ? s1[n] : 0) + (s2 ? s2[n] : 0);
    if (inv_term > 0.0)
    {
      /* the invariant term is not scaled: bring it to the site's scale */
      A += (cnt <= 3) ? ldexp(inv_term, 256 * (int)cnt) : INFINITY;
    }
    if (n >= p->sites)
    {
      const unsigned int k2 = n - p->sites;
      ascA[k2] = A; ascB[k2] = B; ascC[k2] = C; asc_cnt[k2] = cnt;
      continue;
    }
    double w = p->pattern_weights[n];
    double ba = B / A, ca = C / A;
    df -= w * ba;
    ddf += w * (ba * ba - ca);
  }
  if (p->asc_bias_alloc)
  {
    /* d_f, dd_f are derivatives of -lnL */
    double d1, d2;
    asc_derivatives(p, ascA, ascB, ascC, asc_cnt, &d1, &d2);
    df -= d1;
    ddf -= d2;
  }
  *d_f = df;
  *dd_f = ddf;
  free(e0);
  return PLL_SUCCESS;
}

unsigned int pllhip_free_trial_lengths(const pll_partition_t * p) { (void)p; return 1; }

/* several partitions on one tree (include/pllhip.h): the oracle simply walks them, as the reference does
   (src/tree/treeinfo.c:1020-1056) */
int pllhip_update_partials_batch(pll_partition_t * const * partitions, unsigned int partition_count,
                                 const pll_operation_t * ops, unsigned int count)
{
  unsigned int i;
  for (i = 0; i < partition_count; ++i)
    if (partitions[i]) pll_update_partials(partitions[i], ops, count);
  return PLL_SUCCESS;
}

/* several trial branch lengths: the oracle simply repeats the single-length computation
   (include/pllhip.h; the product evaluates them in one pass over the sumtable) */
int pllhip_compute_likelihood_derivatives_multi(pll_partition_t * p, int psc, int csc,
                                                const double * branch_lengths, unsigned int count,
                                                const unsigned int * params_indices,
                                                const double * sumtable, double * d_f, double * dd_f)
{
  unsigned int i;
  for (i = 0; i < count; ++i)
    if (!pll_compute_likelihood_derivatives(p, psc, csc, branch_lengths[i], params_indices, sumtable,
                                            &d_f[i], &dd_f[i]))
      return PLL_FAILURE;
  return PLL_SUCCESS;
}

/* marginal ancestral state probabilities at a node: per site and state,
   sum over rates of w_r pi_i node[n,r,i] * (P other)[n,r,i], normalised */
int pll_compute_node_ancestral(pll_partition_t * p,
                               unsigned int node_clv, int node_sc,
                               unsigned int other_clv, int other_sc,
                               unsigned int matrix_index,
                               const unsigned int * freqs_indices,
                               double * ancestral)
{
  unsigned int S = p->states, Sp = p->states_padded, R = p->rate_cats;
  unsigned int n, r, i, j;
  const double * P = p->pmatrix[matrix_index];
  (void)node_sc; (void)other_sc;
  for (n = 0; n < p->sites; ++n)
  {
    double * out = ancestral + (size_t)n * S;
    double sum = 0.0;
    for (i = 0; i < S; ++i) out[i] = 0.0;
    for (r = 0; r < R; ++r)
    {
      const double * pi = p->frequencies[freqs_indices[r]];
      for (i = 0; i < S; ++i)
      {
        double a = 0.0;
        for (j = 0; j < S; ++j)
          a += P[((size_t)r * S + i) * Sp + j] * orc_clv_at(p, other_clv, n, r, j);
        out[i] += p->rate_weights[r] * pi[i] * orc_clv_at(p, node_clv, n, r, i) * a;
      }
    }
    for (i = 0; i < S; ++i) sum += out[i];
    if (sum > 0.0) for (i = 0; i < S; ++i) out[i] /= sum;
  }
  return PLL_SUCCESS;
}
