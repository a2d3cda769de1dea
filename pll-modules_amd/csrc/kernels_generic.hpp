// kernels_generic.hpp -- kernels that work for any 2 <= S <= 64 and any R.
// They are the correctness baseline of the engine and the production path for
// state counts without a specialised family (kernels_s4.hpp, kernels_s20.hpp).
//
// Layouts: CLV [site][rate][Sp] (states fastest), P-matrix [rate][S][Sp]
// (row = parent state), tip lookup LUT [rate][codes][S] per matrix,
// sumtable [site][rate][Sp].
#pragma once

#include "kernels_common.hpp"
#include "engine.h"
#include <algorithm>

namespace pllhip {

// ---------------------------------------------------------------------------
// P-matrices (+ tip lookup tables as a by-product)
//   P_r(t) = V diag(exp(lambda_k * rate_r * t / (1 - pinv))) V^-1; identity
//   if t == 0.   grid = (count, R), block = 256
//   LUT_r[code][i] = sum_{j in mask(code)} P_r[i][j]
// ---------------------------------------------------------------------------
struct PmatBatch
{
  unsigned midx[MAX_PMAT_PER_LAUNCH];
  double t[MAX_PMAT_PER_LAUNCH];
};

__global__ __launch_bounds__(256) void k_pmatrix(ModelView mv, ParamIdx params, PmatBatch batch,
                                                 unsigned R, double * pmat, double * lut,
                                                 unsigned lut_codes,
                                                 const unsigned long long * tipmap, int staged,
                                                 double * pfrag)
{
  // staged (S*Sp <= 4096):  A[S*Sp] = V diag(exp(lambda rho t)) | B[S*Sp] = V^-1; P overwrites A
  // otherwise:              expk[Sp] | P[S*Sp], operands read from global memory
  extern __shared__ double lds[];
  const unsigned S = mv.S, Sp = mv.Sp;
  const unsigned m = batch.midx[blockIdx.x], r = blockIdx.y;
  const double t = batch.t[blockIdx.x];
  const unsigned pi_ = params.v[r];
  const double * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_), * L = mv.evals(pi_);
  double * P = pmat + ((size_t)m * R + r) * S * Sp;
  double * Pl = staged ? lds : lds + Sp;
  // rows of the result in LDS: Sp + 1 doubles apart when Sp = 64, so that lanes walking down a column (the tip-table
  // build below) meet in different banks instead of one
  const unsigned PS = (staged && Sp == 64) ? 65u : Sp;

  if (t == 0.0)
  {
    for (unsigned e = threadIdx.x; e < S * Sp; e += blockDim.x)
      Pl[(e / Sp) * PS + e % Sp] = (e / Sp == e % Sp) ? 1.0 : 0.0;
  }
  else if (staged)
  {
    const double rt = mv.rates()[r] * t / (1.0 - mv.pinv()[pi_]);
    const unsigned arows = (Sp == 64) ? 64u : S;          // 64 columns: whole 64 x 64 operands, zero beyond S
    double * A = lds, * B = lds + arows * Sp;
    // one exp per eigenvalue (B's first row is free until the loop below fills it), not one per matrix entry
    // every operand load of the thread goes out before anything waits: V and V^-1 (arows * Sp <= 4096 = 16 per
    // thread of the 256) and the eigenvalue -- one memory round trip at the head of the kernel instead of one per
    // operand (or, written as plain loops, one per 256 doubles: kernels_common.hpp, staged_loop)
    double v[16], vi[16];
#pragma unroll
    for (unsigned u = 0; u < 16; ++u)
    {
      const unsigned e = threadIdx.x + u * 256u, ec = e < S * Sp ? e : 0u;
      v[u] = V[ec];
      vi[u] = Vi[ec];
    }
    if (threadIdx.x < S) B[threadIdx.x] = exp(L[threadIdx.x] * rt);
    __syncthreads();
#pragma unroll
    for (unsigned u = 0; u < 16; ++u)
    {
      const unsigned e = threadIdx.x + u * 256u, k = e % Sp;
      if (e < arows * Sp) A[e] = (e < S * Sp && k < S) ? v[u] * B[k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (unsigned u = 0; u < 16; ++u)
    {
      const unsigned e = threadIdx.x + u * 256u;
      if (e < arows * Sp) B[e] = (e < S * Sp) ? vi[u] : 0.0;
    }
    __syncthreads();
    double res[16];
    if (Sp == 64)
    {
      // 64 x 64 operands (61 codon states, any alphabet padded to 64): the product on the matrix
      // cores, wave w = rows 16 w .. 16 w + 15, four 16-column tiles, 16 k-steps.  The scalar loop
      // below takes 36 us of a 55 us launch at 61 states (LDS-bound: two reads per multiply-add).
      // D layout: register v of lane 16 q + n = row 4 v + q, column n of the tile.
      typedef double pm_v4d __attribute__((ext_vector_type(4)));
      const unsigned lane = threadIdx.x & 63u, mt = threadIdx.x >> 6, q = lane >> 4, n = lane & 15u;
      pm_v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      for (unsigned ks = 0; ks < 16; ++ks)
      {
        const double a = A[(16 * mt + n) * Sp + 4 * ks + q];
#pragma unroll
        for (unsigned nt = 0; nt < 4; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[(4 * ks + q) * Sp + 16 * nt + n], acc[nt], 0, 0, 0);
      }
      __syncthreads();
      // straight to the result area (rows >= S are not kept)
#pragma unroll
      for (unsigned nt = 0; nt < 4; ++nt)
#pragma unroll
        for (unsigned v = 0; v < 4; ++v)
        {
          const unsigned i = 16 * mt + 4 * v + q, j = 16 * nt + n;
          if (i < S) Pl[i * PS + j] = (j < S && acc[nt][v] > 0.0) ? acc[nt][v] : 0.0;
        }
    }
    else
    {
#pragma unroll
    for (unsigned u = 0; u < 16; ++u)
    {
      const unsigned e = threadIdx.x + u * 256;
      double acc = 0.0;
      if (e < S * Sp)
      {
        const unsigned i = e / Sp, j = e % Sp;
        if (j < S)
          for (unsigned k = 0; k < S; ++k) acc += A[i * Sp + k] * B[k * Sp + j];
      }
      // a transition probability: the eigen sum leaves ~1e-15 of cancellation noise, which can
      // come out negative where the true entry is smaller than that (three-step codon changes
      // at a slow rate); a negative entry would pass every "all entries < 2^-256" scaling test
      res[u] = acc > 0.0 ? acc : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (unsigned u = 0; u < 16; ++u)
    {
      const unsigned e = threadIdx.x + u * 256;
      if (e < S * Sp) Pl[e] = res[u];
    }
    }
  }
  else
  {
    double * expk = lds;
    const double rt = mv.rates()[r] * t / (1.0 - mv.pinv()[pi_]);
    for (unsigned k = threadIdx.x; k < S; k += blockDim.x) expk[k] = exp(L[k] * rt);
    __syncthreads();
    for (unsigned e = threadIdx.x; e < S * Sp; e += blockDim.x)
    {
      const unsigned i = e / Sp, j = e % Sp;
      double acc = 0.0;
      if (j < S)
        for (unsigned k = 0; k < S; ++k) acc += V[i * Sp + k] * expk[k] * Vi[k * Sp + j];
      Pl[e] = acc > 0.0 ? acc : 0.0;
    }
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < S * Sp; e += blockDim.x) P[e] = Pl[(e / Sp) * PS + e % Sp];
  if (pfrag)
  {
    // 20 states: the matrix once more as compact MFMA A fragments (kernels_s20.hpp, s20_fill_cfrags),
    // so that the chain kernel fills its LDS with a plain copy
    double * F = pfrag + ((size_t)m * R + r) * 400;
    for (unsigned x = threadIdx.x; x < 400; x += blockDim.x)
    {
      // (kernels_s20.hpp, s20_cfrag_index: [k-step][row group][q][row in group])
      const unsigned t = x >> 4, ks = t / 5, g = t - 5 * ks;
      const unsigned i = 4 * g + (x & 3u), j = 4 * ks + ((x >> 2) & 3u);
      F[x] = Pl[i * PS + j];
    }
  }
  if (lut)
  {
    double * T = lut + ((size_t)m * R + r) * lut_codes * S;
    staged_loop<16>(lut_codes * S, [=](unsigned e) { return tipmap[e / S]; }, [=](unsigned e, unsigned long long mask)
    {
      const unsigned i = e % S;
      double acc = 0.0;
      if (mask && !(mask & (mask - 1)))               // one state: a column of P
        acc = Pl[i * PS + (unsigned)__ffsll((long long)mask) - 1];
      else
        for (unsigned j = 0; j < S; ++j)
          if ((mask >> j) & 1ULL) acc += Pl[i * PS + j];
      T[e] = acc;
    });
  }
}

// rebuild every LUT from the stored P-matrices (after the tip code table grew)
__global__ __launch_bounds__(256) void k_rebuild_lut(unsigned S, unsigned Sp, unsigned R,
                                                     const double * pmat, double * lut,
                                                     unsigned lut_codes,
                                                     const unsigned long long * tipmap)
{
  const unsigned m = blockIdx.x, r = blockIdx.y;
  const double * P = pmat + ((size_t)m * R + r) * S * Sp;
  double * T = lut + ((size_t)m * R + r) * lut_codes * S;
  for (unsigned e = threadIdx.x; e < lut_codes * S; e += blockDim.x)
  {
    const unsigned c = e / S, i = e % S;
    const unsigned long long mask = tipmap[c];
    double s = 0.0;
    for (unsigned j = 0; j < S; ++j)
      if ((mask >> j) & 1ULL) s += P[i * Sp + j];
    T[e] = s;
  }
}

// ---------------------------------------------------------------------------
// partials, generic: one thread per output element (site, rate, state).
//   parent[n,r,i] = (sum_j P1[r,i,j] c1[n,r,j]) * (sum_j P2[r,i,j] c2[n,r,j])
//   a coded tip child contributes LUT[r][code[n]][i] instead of the dot product.
//   scaling: block owns whole sites; LDS flag per site.
// grid = (site groups, ops), block = 256; a block processes `spb` sites per
// step, spb * R * Sp <= 256 * GEN_MAX_ELEMS.
// ---------------------------------------------------------------------------
constexpr unsigned GEN_MAX_ELEMS = 4;

__global__ __launch_bounds__(256) void k_partials_generic(OpBatch batch, unsigned N, unsigned R,
                                                          unsigned S, unsigned Sp,
                                                          unsigned lut_codes, unsigned spb, unsigned rate_scalers)
{
  // big[s] != 0: site s has an entry >= threshold; per-rate scalers: big[s * R + r] for (site, rate)
  __shared__ int big[256];
  const unsigned G = rate_scalers ? R : 1u;      // scaling groups per site
  const OpDesc & op = batch.op[blockIdx.y];
  const unsigned E = R * Sp;               // elements per site
  const unsigned per_step = spb * E;

  for (unsigned long long n0 = (unsigned long long)blockIdx.x * spb; n0 < N;
       n0 += (unsigned long long)gridDim.x * spb)
  {
    if (threadIdx.x < spb * G) big[threadIdx.x] = 0;
    __syncthreads();
    double val[GEN_MAX_ELEMS];
#pragma unroll
    for (unsigned q = 0; q < GEN_MAX_ELEMS; ++q)
    {
      const unsigned e = threadIdx.x + q * 256;
      val[q] = 0.0;
      if (e >= per_step) continue;
      const unsigned s = e / E, rem = e % E, r = rem / Sp, i = rem % Sp;
      const unsigned long long n = n0 + s;
      if (n >= N || i >= S) continue;
      double a, b;
      if (op.codes1)
        a = op.lut1[((size_t)r * lut_codes + op.codes1[n]) * S + i];
      else
      {
        const double * c = op.clv1 + (n * R + r) * Sp;
        const double * row = op.pmat1 + ((size_t)r * S + i) * Sp;
        a = 0.0;
        for (unsigned j = 0; j < S; ++j) a += row[j] * c[j];
      }
      if (op.codes2)
        b = op.lut2[((size_t)r * lut_codes + op.codes2[n]) * S + i];
      else
      {
        const double * c = op.clv2 + (n * R + r) * Sp;
        const double * row = op.pmat2 + ((size_t)r * S + i) * Sp;
        b = 0.0;
        for (unsigned j = 0; j < S; ++j) b += row[j] * c[j];
      }
      val[q] = a * b;
      if (!(val[q] < SCALE_THRESHOLD)) big[rate_scalers ? s * R + r : s] = 1;   // benign race: all writers store 1
    }
    __syncthreads();
#pragma unroll
    for (unsigned q = 0; q < GEN_MAX_ELEMS; ++q)
    {
      const unsigned e = threadIdx.x + q * 256;
      if (e >= per_step) continue;
      const unsigned s = e / E, rem = e % E;
      const unsigned long long n = n0 + s;
      if (n >= N) continue;
      const unsigned r = rem / Sp;
      const bool rescale = op.parent_scaler && !big[rate_scalers ? s * R + r : s];
      op.parent[n * E + rem] = rescale ? val[q] * SCALE_FACTOR : val[q];
      if (op.parent_scaler && (rate_scalers ? rem % Sp == 0 : rem == 0))
      {
        const unsigned long long sx = rate_scalers ? n * R + r : n;
        unsigned cnt = (op.scaler1 ? op.scaler1[sx] : 0u) + (op.scaler2 ? op.scaler2[sx] : 0u);
        op.parent_scaler[sx] = cnt + (rescale ? 1u : 0u);
      }
    }
    __syncthreads();
  }
}

// value of state j at (site n, rate r) for a full CLV or a coded tip
__device__ inline double node_value(const NodeRef & nd, const unsigned long long * tipmap,
                                    unsigned long long n, unsigned r, unsigned j,
                                    unsigned R, unsigned Sp)
{
  if (nd.codes) return (double)((tipmap[nd.codes[n]] >> j) & 1ULL);
  return nd.clv[(n * R + r) * Sp + j];
}

// ---------------------------------------------------------------------------
// edge / root log-likelihood, generic: one thread per site, fixed-order block
// reduction, one partial per block.  grid <= REDUCE_BLOCKS, block = 256.
//   L_n = sum_r w_r [ (1-pinv) sum_i pi_i p[n,r,i] sum_j P[r,i,j] c[n,r,j]
//                     + pinv * pi[inv(n)] ]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_lnl_generic(ModelView mv, ParamIdx freqs_idx,
                                                          NodeRef parent, NodeRef child,
                                                          const double * pmat,   // nullptr: root lnL
                                                          const double * lut, unsigned lut_codes,
                                                          const unsigned * ps, const unsigned * cs,
                                                          const unsigned * weights,
                                                          const int * invariant,
                                                          const unsigned long long * tipmap,
                                                          unsigned N, unsigned R,
                                                          double * persite, ReduceOut block_out,
                                                          unsigned rate_scalers)
{
  __shared__ double scratch[4];
  const unsigned S = mv.S, Sp = mv.Sp;
  double acc = 0.0;
  for (unsigned long long n = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; n < N;
       n += (unsigned long long)gridDim.x * blockDim.x)
  {
    double site = 0.0, inv = 0.0;
    const unsigned min_cnt = rate_scalers ? rate_min_count(ps, cs, n, R) : 0u;
    for (unsigned r = 0; r < R; ++r)
    {
      const unsigned fi = freqs_idx.v[r];
      const double * pi = mv.freqs(fi);
      const double pinv = mv.pinv()[fi];
      double lr = 0.0;
      for (unsigned i = 0; i < S; ++i)
      {
        double a;
        if (!pmat)
          a = 1.0;
        else if (child.codes)
          a = lut[((size_t)r * lut_codes + child.codes[n]) * S + i];
        else
        {
          const double * row = pmat + ((size_t)r * S + i) * Sp;
          const double * c = child.clv + (n * R + r) * Sp;
          a = 0.0;
          for (unsigned j = 0; j < S; ++j) a += row[j] * c[j];
        }
        lr += pi[i] * node_value(parent, tipmap, n, r, i, R, Sp) * a;
      }
      if (rate_scalers) lr *= rate_factor(ps, cs, n, R, r, min_cnt);
      const double w = mv.weights()[r];
      if (pinv > 0.0)
      {
        site += w * (1.0 - pinv) * lr;
        if (invariant && invariant[n] >= 0) inv += w * pinv * pi[invariant[n]];
      }
      else
        site += w * lr;
    }
    const unsigned cnt = rate_scalers ? min_cnt : (ps ? ps[n] : 0u) + (cs ? cs[n] : 0u);
    const double l = site_loglh(site, cnt, inv);
    if (persite) persite[n] = l;
    acc += l * (double)weights[n];
  }
  const double tot = block_sum_256(acc, scratch);
  grid_reduce_finish1(tot, block_out, scratch);
}

// ---------------------------------------------------------------------------
// sumtable, generic: one thread per (site, rate, k)
//   sum[n,r,k] = (sum_i p[n,r,i] pi_i V[i,k]) * (sum_j V^-1[k,j] c[n,r,j])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumtable_generic(ModelView mv, ParamIdx params,
                                                          NodeRef parent, NodeRef child,
                                                          const unsigned long long * tipmap,
                                                          unsigned N, unsigned R, double * sumtable)
{
  const unsigned S = mv.S, Sp = mv.Sp;
  const unsigned long long total = (unsigned long long)N * R * Sp;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * blockDim.x)
  {
    const unsigned k = e % Sp;
    const unsigned long long nr = e / Sp;
    const unsigned r = nr % R;
    const unsigned long long n = nr / R;
    double out = 0.0;
    if (k < S)
    {
      const unsigned pi_ = params.v[r];
      const double * pi = mv.freqs(pi_), * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_);
      double a = 0.0, b = 0.0;
      for (unsigned i = 0; i < S; ++i)
      {
        a += node_value(parent, tipmap, n, r, i, R, Sp) * pi[i] * V[i * Sp + k];
        b += Vi[k * Sp + i] * node_value(child, tipmap, n, r, i, R, Sp);
      }
      out = a * b;
    }
    sumtable[e] = out;
  }
}

// ---------------------------------------------------------------------------
// derivatives of -lnL at K trial branch lengths from ONE pass over the sumtable,
// generic: one thread per site; coefficient tables
// e0/e1/e2[j][r][k] = w_r' exp(l t_j) {1, l, l^2}, l = lambda_k rate_r/(1-pinv),
// are built per block in LDS.   totals: df[0], ddf[0], df[1], ddf[1], ...
// Every trial length goes through the arithmetic of the K = 1 instance, so its
// result does not depend on which other lengths share the launch.
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_derivatives_generic(ModelView mv, ParamIdx params, TrialLengths tl,
                                                             const double * sumtable,
                                                             const unsigned * ps, const unsigned * cs,
                                                             const unsigned * weights,
                                                             const int * invariant,
                                                             unsigned N, unsigned R,
                                                             ReduceOut block_out, unsigned rate_scalers)
{
  extern __shared__ double lds[];          // per trial length: e0 | e1 | e2, each R*S
  __shared__ double scratch[4];
  const unsigned S = mv.S, Sp = mv.Sp, RS = R * S;
  for (unsigned x = threadIdx.x; x < K * RS; x += blockDim.x)
  {
    const unsigned j = x / RS, q = x % RS;
    const unsigned r = q / S, k = q % S, pi_ = params.v[r];
    const double pinv = mv.pinv()[pi_];
    const double lam = mv.evals(pi_)[k] * mv.rates()[r] / (1.0 - pinv);
    const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
    const double ex = exp(lam * tl.t[j]);
    double * e = lds + (size_t)j * 3 * RS;
    e[q] = wr * ex;
    e[RS + q] = wr * ex * lam;
    e[2 * RS + q] = wr * ex * lam * lam;
  }
  __syncthreads();

  double df[K], ddf[K];
#pragma unroll
  for (int j = 0; j < K; ++j) df[j] = ddf[j] = 0.0;
  for (unsigned long long n = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; n < N;
       n += (unsigned long long)gridDim.x * blockDim.x)
  {
    double inv = 0.0;
    for (unsigned r = 0; r < R; ++r)
    {
      const unsigned pi_ = params.v[r];
      const double pinv = mv.pinv()[pi_];
      if (pinv > 0.0 && invariant && invariant[n] >= 0)
        inv += mv.weights()[r] * pinv * mv.freqs(pi_)[invariant[n]];
    }
    const unsigned min_cnt = rate_scalers ? rate_min_count(ps, cs, n, R) : 0u;
    if (inv > 0.0)
    {
      const unsigned cnt = rate_scalers ? min_cnt : (ps ? ps[n] : 0u) + (cs ? cs[n] : 0u);
      inv = (cnt <= 3) ? ldexp(inv, 256 * (int)cnt) : INFINITY;
    }
    const double w = (double)weights[n];
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
      const double * e0 = lds + (size_t)j * 3 * RS, * e1 = e0 + RS, * e2 = e1 + RS;
      double A = 0.0, B = 0.0, C = 0.0;
      for (unsigned r = 0; r < R; ++r)
      {
        const double * st = sumtable + (n * R + r) * Sp;
        if (rate_scalers)
        {
          // the rate's sums first, then the factor that brings it to the site's smallest count
          double a = 0.0, b = 0.0, c = 0.0;
          for (unsigned k = 0; k < S; ++k)
          {
            const double v = st[k];
            a += v * e0[r * S + k];
            b += v * e1[r * S + k];
            c += v * e2[r * S + k];
          }
          const double f = rate_factor(ps, cs, n, R, r, min_cnt);
          A += f * a;
          B += f * b;
          C += f * c;
          continue;
        }
        for (unsigned k = 0; k < S; ++k)
        {
          const double v = st[k];
          A += v * e0[r * S + k];
          B += v * e1[r * S + k];
          C += v * e2[r * S + k];
        }
      }
      if (inv > 0.0) A += inv;
      const double ba = B / A, ca = C / A;
      df[j] -= w * ba;
      ddf[j] += w * (ba * ba - ca);
    }
  }
  double tot[2 * K];
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    tot[2 * j] = block_sum_256(df[j], scratch);
    tot[2 * j + 1] = block_sum_256(ddf[j], scratch);
  }
  grid_reduce_finish<2 * K>(tot, block_out, scratch);
}

// ---------------------------------------------------------------------------
// invariant sites: inv[n] = lowest state compatible with every tip, else -1
// ---------------------------------------------------------------------------
struct TipTable { const double * const * clv; const uint8_t * const * codes; };

__global__ __launch_bounds__(256) void k_invariant(const double * const * tip_clv,
                                                   const uint8_t * const * tip_codes,
                                                   const unsigned long long * tipmap,
                                                   unsigned tips, unsigned N, unsigned R,
                                                   unsigned S, unsigned Sp, unsigned brows,
                                                   int * invariant)
{
  for (unsigned long long n = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; n < N;
       n += (unsigned long long)gridDim.x * blockDim.x)
  {
    unsigned long long common = ~0ULL;
    for (unsigned t = 0; t < tips; ++t)
    {
      unsigned long long m = 0;
      if (tip_codes[t])
        m = tipmap[tip_codes[t][n]];
      else
      {
        for (unsigned j = 0; j < S; ++j)
        {
          // rate 0 of site n: API layout, or the 32-site blocked layout of kernels_s20.hpp
          const double v = brows ? tip_clv[t][(((n >> 5) * R) * brows + j) * 32 + (n & 31)]
                                 : tip_clv[t][n * R * Sp + j];
          if (v > 0.0) m |= (1ULL << j);
        }
      }
      common &= m;
    }
    if (S < 64) common &= ((1ULL << S) - 1ULL);
    invariant[n] = common ? (int)__builtin_ctzll(common) : -1;
  }
}

// ---------------------------------------------------------------------------
// marginal ancestral state probabilities at a node (src/tree/treeinfo.c:1698):
//   anc[n][i] ~ sum_r w_r pi_i node[n,r,i] * sum_j P[r,i,j] other[n,r,j], normalised
// one thread per site; works on the API layout and on the 32-site blocked layout
// (element (n, r, j) at ((n/32 * R + r) * brows + j) * 32 + n%32, brows = state rows per unit)
// ---------------------------------------------------------------------------
__device__ inline double clv_elem(const NodeRef & nd, const unsigned long long * tipmap, unsigned brows,
                                  unsigned long long n, unsigned r, unsigned j, unsigned R, unsigned Sp)
{
  if (nd.codes) return (double)((tipmap[nd.codes[n]] >> j) & 1ULL);
  return brows ? nd.clv[(((n >> 5) * R + r) * brows + j) * 32 + (n & 31)]
               : nd.clv[(n * R + r) * Sp + j];
}

__global__ __launch_bounds__(256) void k_node_ancestral(ModelView mv, ParamIdx fidx, NodeRef node,
                                                        NodeRef other, const double * pmat,
                                                        const unsigned long long * tipmap, unsigned brows,
                                                        unsigned N, unsigned R, double * out)
{
  const unsigned S = mv.S, Sp = mv.Sp;
  for (unsigned long long n = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; n < N;
       n += (unsigned long long)gridDim.x * blockDim.x)
  {
    double * o = out + n * S;
    double sum = 0.0;
    for (unsigned i = 0; i < S; ++i)
    {
      double v = 0.0;
      for (unsigned r = 0; r < R; ++r)
      {
        const double * row = pmat + ((size_t)r * S + i) * Sp;
        double a = 0.0;
        for (unsigned j = 0; j < S; ++j) a += row[j] * clv_elem(other, tipmap, brows, n, r, j, R, Sp);
        v += mv.weights()[r] * mv.freqs(fidx.v[r])[i] * clv_elem(node, tipmap, brows, n, r, i, R, Sp) * a;
      }
      o[i] = v;
      sum += v;
    }
    if (sum > 0.0)
      for (unsigned i = 0; i < S; ++i) o[i] /= sum;
  }
}

// ---------------------------------------------------------------------------
// ascertainment-bias correction (PLL_ATTRIB_AB_*): the S constant patterns sit behind the
// alignment in every per-site array; their share of the likelihood is a closed-form correction
// the host applies to the sums over the alignment.  Two small kernels hand it what it needs.
// ---------------------------------------------------------------------------
// log-likelihoods of the constant patterns (the tail of the per-site output of the lnL kernel
// that ran before on the same stream) -> mapped host memory, then the sequence word
__global__ void k_publish_tail(const double * tail, double * dst, unsigned n, unsigned long long * flag,
                               unsigned long long seq)
{
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = tail[i];
  __syncthreads();
  if (threadIdx.x == 0 && flag)
  {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// A, B, C (likelihood and its first two derivatives by the branch length) and the
// scaling count of every constant pattern at `ntrial` branch lengths, from the sumtable rows
// behind the alignment: out[((j * S + k) * 4 + {0, 1, 2, 3}] (mapped host memory).
// One block; thread = (trial length, pattern).  brows: state rows per blocked unit (0: API layout).
__global__ __launch_bounds__(256) void k_asc_abc(ModelView mv, ParamIdx params, TrialLengths tl, unsigned ntrial,
                                                 const double * sumtable, const unsigned * ps, const unsigned * cs,
                                                 unsigned Nreal, unsigned R, unsigned brows, unsigned rate_scalers,
                                                 double * out)
{
  const unsigned S = mv.S, Sp = mv.Sp;
  for (unsigned x = threadIdx.x; x < ntrial * S; x += blockDim.x)
  {
    const unsigned j = x / S, k0 = x % S;
    const unsigned long long n = (unsigned long long)Nreal + k0;
    const unsigned min_cnt = rate_scalers ? rate_min_count(ps, cs, n, R) : 0u;
    double A = 0.0, B = 0.0, C = 0.0, inv = 0.0;
    for (unsigned r = 0; r < R; ++r)
    {
      const unsigned pi_ = params.v[r];
      const double pinv = mv.pinv()[pi_];
      const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
      double a = 0.0, b = 0.0, c = 0.0;
      for (unsigned k = 0; k < S; ++k)
      {
        const double v = brows ? sumtable[(((n >> 5) * R + r) * brows + k) * 32 + (n & 31)]
                               : sumtable[(n * R + r) * Sp + k];
        const double lam = mv.evals(pi_)[k] * mv.rates()[r] / (1.0 - pinv);
        const double e = wr * exp(lam * tl.t[j]);
        a += v * e;
        b += v * e * lam;
        c += v * e * lam * lam;
      }
      const double f = rate_scalers ? rate_factor(ps, cs, n, R, r, min_cnt) : 1.0;
      A += f * a; B += f * b; C += f * c;
      if (pinv > 0.0) inv += mv.weights()[r] * pinv * mv.freqs(pi_)[k0];      // a constant pattern IS invariant
    }
    const unsigned cnt = rate_scalers ? min_cnt : (ps ? ps[n] : 0u) + (cs ? cs[n] : 0u);
    if (inv > 0.0) A += (cnt <= 3) ? ldexp(inv, 256 * (int)cnt) : INFINITY;
    out[(size_t)x * 4] = A;
    out[(size_t)x * 4 + 1] = B;
    out[(size_t)x * 4 + 2] = C;
    out[(size_t)x * 4 + 3] = (double)cnt;
  }
}

// expand a coded tip into a 0/1 CLV (host materialisation path)
__global__ __launch_bounds__(256) void k_expand_codes(const uint8_t * codes,
                                                      const unsigned long long * tipmap,
                                                      unsigned N, unsigned R, unsigned S, unsigned Sp,
                                                      double * out)
{
  const unsigned long long total = (unsigned long long)N * R * Sp;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * blockDim.x)
  {
    const unsigned j = e % Sp;
    const unsigned long long n = e / Sp / R;
    out[e] = (j < S) ? (double)((tipmap[codes[n]] >> j) & 1ULL) : 0.0;
  }
}

// --- launchers -------------------------------------------------------------

static int launch_partials_generic(Engine * e, const OpBatch & batch, unsigned nops)
{
  const unsigned E = e->R * e->Sp;
  const unsigned spb = std::max(1u, 256u / E);
  const unsigned long long groups = ((unsigned long long)e->N + spb - 1) / spb;
  const unsigned gx = (unsigned)std::max<unsigned long long>(
      1, std::min<unsigned long long>(groups, e->cu_count * 16ULL));
  hipLaunchKernelGGL(k_partials_generic, dim3(gx, nops), dim3(256), 0, e->stream,
                     batch, e->N, e->R, e->S, e->Sp, e->lut_codes, spb, e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_edge_lnl_generic(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                                   const NodeRef & parent, const NodeRef & child,
                                   const double * pm, const double * lut,
                                   const unsigned * ps, const unsigned * cs,
                                   double * persite, unsigned nblocks)
{
  hipLaunchKernelGGL(k_edge_lnl_generic, dim3(nblocks), dim3(256), 0, e->stream,
                     mv, fidx, parent, child, pm, lut, e->lut_codes, ps, cs,
                     e->d_weights, e->d_invariant, e->d_tipmap, e->N, e->R, persite, reduce_out(e),
                     e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_sumtable_generic(Engine * e, const ModelView & mv, const ParamIdx & params,
                                   const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  const unsigned long long total = (unsigned long long)e->N * e->R * e->Sp;
  const unsigned gx = (unsigned)std::max<unsigned long long>(
      1, std::min<unsigned long long>((total + 255) / 256, e->cu_count * 32ULL));
  hipLaunchKernelGGL(k_sumtable_generic, dim3(gx), dim3(256), 0, e->stream,
                     mv, params, parent, child, e->d_tipmap, e->N, e->R, d_sum);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// K trial lengths -> the smallest instance that holds them (1, 2, 4, 8); spare slots repeat the last length
#define PLLHIP_DISPATCH_K(count, CALL) \
  do { if ((count) <= 1) { CALL(1); } else if ((count) == 2) { CALL(2); } \
       else if ((count) <= 4) { CALL(4); } else { CALL(8); } } while (0)

inline unsigned trial_instance(unsigned count) { return count <= 1 ? 1u : count == 2 ? 2u : count <= 4 ? 4u : 8u; }

static int launch_derivatives_generic(Engine * e, const ModelView & mv, const ParamIdx & params,
                                      const TrialLengths & tl, unsigned count, const double * d_sum,
                                      const unsigned * ps, const unsigned * cs, unsigned nblocks)
{
#define PLLHIP_CALL(KK) \
  hipLaunchKernelGGL(k_derivatives_generic<KK>, dim3(nblocks), dim3(256), sizeof(double) * 3 * KK * e->R * e->S, \
                     e->stream, mv, params, tl, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->R, reduce_out(e), \
                     e->rate_scalers ? 1u : 0u)
  PLLHIP_DISPATCH_K(count, PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

} // namespace pllhip
