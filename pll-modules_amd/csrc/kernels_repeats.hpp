// kernels_repeats.hpp -- site repeats, first step: cherries (PLL_ATTRIB_SITE_REPEATS).
//
// libpll-2's site repeats compute the vector of a node once per CLASS of sites that share the pattern of the
// subtree below it (the reference selects the attribute in its test harness, test/src/common.c:31, and carries
// the per-node class tables through its checkpoints, src/binary/binary_io_operations.c:231-282).  For a cherry --
// a tip x tip operation -- the class of a site is simply its pair of tip codes: at most codes^2 classes (a few
// hundred) however long the alignment is.  With the attribute set the 20-state family (and, with scalar kernels of
// the same structure at the end of this file, the 4-state family) therefore
//   * computes a cherry's vector per code pair (k_cherry_build: the arithmetic of the tip x tip operation on a
//     "pseudo alignment" whose sites are the pairs; a few hundred KiB that stay in the caches) and writes per site
//     only the 16-bit class code and the scaler count (k_cherry_sites: 6 B instead of 640 B per site);
//   * hands the cherry to the operation above it as a "wide tip": a child read through a lookup table with one row
//     per class, [rate][class][20] = P . vector(class), built per traversal on the matrix cores with the very MFMA
//     sequence an inner child takes (k_pair_lut) -- so the operation's result is, bit for bit, what it computes
//     from the expanded vector;
//   * expands the vector to the site-indexed form only for a reader that needs it (k_cherry_expand: an edge lnL
//     or a sumtable AT the cherry, pllhip_get_clv, a checkpoint).
// What a caller can observe -- vectors, scaler counts, likelihoods, derivatives -- is identical to the attribute
// being off (tests/test_site_repeats.py).  Deeper classes (tip x cherry, ...) are the next step.
#pragma once

#include "kernels_common.hpp"
#include "kernels_s20.hpp"
#include "engine.h"

namespace pllhip {

// one virtual cherry of a traversal (device memory, part of the resident schedule)
struct CherryJob
{
  const double * lut1, * lut2;       // the two tips' lookup tables [rate][lut_codes][20]
  const uint8_t * codes1, * codes2;
  double * table;                    // blocked pseudo-CLV [pair block][rate][unit]
  uint8_t * flags;                   // per pair: the vector was scaled
  unsigned short * pair;             // per site: class code
  unsigned * parent_scaler;          // per site, or null
};

// one lookup table of a traversal: table(cherry) seen through the P-matrix of the consumer's branch
struct PairLutJob
{
  const double * table;
  const double * pfrag;              // compact A fragments [rate][400] of the branch's matrices
  double * out;                      // [rate][pairs][20]
};

// the cherry per code pair.  grid = (pair blocks / 4, jobs), block = 256 (a wave per 32-pair block)
template <unsigned RT>
__global__ __launch_bounds__(256) void k_cherry_build(const CherryJob * jobs, unsigned lut_codes, unsigned ncodes)
{
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npairs = ncodes * ncodes, npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * lut1 = as_global(job.lut1), * lut2 = as_global(job.lut2);
  double * table = as_global(job.table);
  uint8_t * flags = as_global(job.flags);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
  // pairs beyond the table repeat pair 0: their columns are never read
  const unsigned ae = pe < npairs ? pe / ncodes : 0, be = pe < npairs ? pe % ncodes : 0;
  const unsigned ao = po < npairs ? po / ncodes : 0, bo = po < npairs ? po % ncodes : 0;
  double2 X[RT][5];
  int small_e = 1, small_o = 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    double2 t2[5];
    s20_child_tip(lut1 + (size_t)r * lut_codes * 20, ae, ao, q, X[r]);
    s20_child_tip(lut2 + (size_t)r * lut_codes * 20, be, bo, q, t2);
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      X[r][k].x *= t2[k].x;
      X[r][k].y *= t2[k].y;
      small_e &= (X[r][k].x < SCALE_THRESHOLD);
      small_o &= (X[r][k].y < SCALE_THRESHOLD);
    }
  }
  double fe = 1.0, fo = 1.0;
  if (job.parent_scaler)
  {
    small_e = s20_and_q(small_e);
    small_o = s20_and_q(small_o);
    fe = small_e ? SCALE_FACTOR : 1.0;
    fo = small_o ? SCALE_FACTOR : 1.0;
    if (q == 0)
    {
      if (pe < npairs) flags[pe] = (uint8_t)small_e;
      if (po < npairs) flags[po] = (uint8_t)small_o;
    }
  }
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
#pragma unroll
    for (int k = 0; k < 5; ++k) { X[r][k].x *= fe; X[r][k].y *= fo; }
    s20_store_d(table + ((size_t)blk * RT + r) * S20_UNIT, lane, X[r]);
  }
}

// class code and scaler count per site.  grid = (chunks, jobs), block = 256
__global__ __launch_bounds__(256) void k_cherry_sites(const CherryJob * jobs, unsigned ncodes, unsigned nalloc)
{
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const uint8_t * c1 = as_global(job.codes1), * c2 = as_global(job.codes2);
  const uint8_t * flags = as_global(job.flags);
  unsigned short * pair = as_global(job.pair);
  unsigned * ps = as_global(job.parent_scaler);
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < nalloc; s += gridDim.x * 256u)
  {
    const unsigned a = c1[s], b = c2[s];
    const unsigned p = (a < ncodes && b < ncodes) ? a * ncodes + b : 0u;       // (padding sites carry code 0)
    pair[s] = (unsigned short)p;
    if (ps) ps[s] = flags[p];
  }
}

// lookup tables of the wide tips of a traversal.  grid = (pair blocks / 4, jobs), block = 256,
// dynamic LDS = RT * 400 doubles (the compact fragments of the branch's matrices)
template <unsigned RT>
__global__ __launch_bounds__(256) void k_pair_lut(const PairLutJob * jobs, unsigned npairs)
{
  extern __shared__ double cfrag[];
  const PairLutJob job = plan_fetch(jobs + blockIdx.y);
  const double * pf = as_global(job.pfrag);
  staged_copy<8>(cfrag, pf, RT * S20_CFRAGS);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * table = as_global(job.table);
  double * out = as_global(job.out);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    double2 t[5];
    s20_child_inner_c(table + ((size_t)blk * RT + r) * S20_UNIT, cfrag + r * S20_CFRAGS, lane, t, false);
#pragma unroll
    for (unsigned k = 0; k < 5; ++k)
    {
      const unsigned i = s20_row(k, q);
      if (pe < npairs) out[((size_t)r * npairs + pe) * 20 + i] = t[k].x;
      if (po < npairs) out[((size_t)r * npairs + po) * 20 + i] = t[k].y;
    }
  }
}

// the site-indexed vector of a cherry, for a reader that needs it.  grid = site blocks / 4, block = 256
__global__ __launch_bounds__(256) void k_cherry_expand(const double * table, const unsigned short * pair,
                                                       unsigned nblk, unsigned R, double * clv)
{
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    const unsigned pe = pair[site0], po = pair[site0 + 1];
    for (unsigned r = 0; r < R; ++r)
    {
      const double * ue = table + ((size_t)(pe / S20_BS) * R + r) * S20_UNIT + (q * 16 + (pe % S20_BS) / 2) * 2 + (pe & 1u);
      const double * uo = table + ((size_t)(po / S20_BS) * R + r) * S20_UNIT + (q * 16 + (po % S20_BS) / 2) * 2 + (po & 1u);
      double2 t[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] = make_double2(ue[k * 128], uo[k * 128]);
      s20_store_d(clv + ((size_t)blk * R + r) * S20_UNIT, lane, t);
    }
  }
}

// ---------------------------------------------------------------------------
// the same for the 4-state family (kernels_s4.hpp; vectors in the API layout [site][rate][4], tip tables
// [rate][16 codes][4]): at most 256 classes per cherry
// ---------------------------------------------------------------------------
// grid = (ceil(pairs / 256), jobs), block = 256: a thread per class
__global__ __launch_bounds__(256) void k_cherry_build_s4(const CherryJob * jobs, unsigned R, unsigned ncodes)
{
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned npairs = ncodes * ncodes;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= npairs) return;
  const double * lut1 = as_global(job.lut1), * lut2 = as_global(job.lut2);
  double * table = as_global(job.table);
  const unsigned a = p / ncodes, b = p % ncodes;
  bool small = true;
  for (unsigned r = 0; r < R; ++r)
    for (unsigned i = 0; i < 4; ++i)
    {
      const double v = lut1[(r * 16 + a) * 4 + i] * lut2[(r * 16 + b) * 4 + i];
      table[((size_t)p * R + r) * 4 + i] = v;
      small = small && (v < SCALE_THRESHOLD);
    }
  if (job.parent_scaler)
  {
    as_global(job.flags)[p] = small ? 1 : 0;
    if (small)
      for (unsigned e = 0; e < R * 4; ++e) table[(size_t)p * R * 4 + e] *= SCALE_FACTOR;
  }
}

// [rate][class][4] = P(rate) . vector(class), with the association of s4_half_matvec:
// (P_i0 x_0 + P_i1 x_1) + (P_i2 x_2 + P_i3 x_3).  grid = (ceil(pairs * R / 256), jobs), block = 256;
// job.pfrag holds the branch's matrices [rate][4][4]
__global__ __launch_bounds__(256) void k_pair_lut_s4(const PairLutJob * jobs, unsigned npairs, unsigned R)
{
  const PairLutJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned x = blockIdx.x * 256u + threadIdx.x;
  if (x >= npairs * R) return;
  const unsigned p = x / R, r = x % R;
  const double * P = as_global(job.pfrag) + r * 16;
  const double * v = as_global(job.table) + ((size_t)p * R + r) * 4;
  double * out = as_global(job.out) + ((size_t)r * npairs + p) * 4;
  for (unsigned i = 0; i < 4; ++i)
  {
    const double lo = P[i * 4 + 0] * v[0] + P[i * 4 + 1] * v[1];
    const double hi = P[i * 4 + 2] * v[2] + P[i * 4 + 3] * v[3];
    out[i] = (i < 2) ? lo + hi : hi + lo;
  }
}

// the site-indexed vector of a cherry: grid-stride over (site, rate) columns
__global__ __launch_bounds__(256) void k_cherry_expand_s4(const double * table, const unsigned short * pair,
                                                          unsigned N, unsigned R, double * clv)
{
  const unsigned long long total = (unsigned long long)N * R;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 256u + threadIdx.x; c < total;
       c += (unsigned long long)gridDim.x * 256u)
  {
    const unsigned long long n = c / R;
    const unsigned r = (unsigned)(c % R);
    const double * src = table + ((size_t)pair[n] * R + r) * 4;
    double * dst = clv + c * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  }
}

} // namespace pllhip
