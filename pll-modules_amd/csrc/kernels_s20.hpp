// kernels_s20.hpp -- 20-state (protein) kernel family on the fp64 matrix cores.
//
// The P * CLV product of a 20-state partition is a dense [20x20] x [20 x sites]
// contraction per rate category: the one place on this path where MFMA is a
// genuine fit (v_mfma_f64_16x16x4_f64; 16 FMAs per lane for one A and one B
// double per lane, so operand delivery -- not flops -- stops being the limit).
//
// Device layout ("blocked", private to this family; the API layout
// [site][rate][state] only exists in host mirrors):
//     clv[site_block b][rate r][state j][site-in-block s],  32 sites per block
//   i.e. one (block, rate) *unit* is a 20 x 32 fp64 matrix (5 KiB), state-major.
//   With lane l = 16*q + n (q = l>>4, n = l&15) holding the two sites 2n, 2n+1:
//     * MFMA B operand, k-step ks:  rows j = 4*ks + q  -> address (j*32 + 2n)*8
//       = ks*1024 + l*16 bytes: ONE fully coalesced global_load_dwordx4 (1 KiB
//       per wave instruction) feeds two MFMAs (even / odd sites);
//     * MFMA D result, register v: rows i = q + 4*v    -> same address form:
//       ONE fully coalesced 1 KiB store per register.
//   What one operation stores is bit-for-bit what the next one loads: no LDS
//   staging, no transposes, no bank conflicts for CLV data at all.  LDS holds
//   only the P-matrices, pre-arranged as per-lane A fragments.
//
// Work decomposition: one wavefront owns a site block and walks its R rates
// (a wave = one site-block x rate unit at a time); waves never synchronise.
// The per-site scaling vote spans all R*20 entries of a site, which the wave
// sees after its last rate; the (rare) rescale is a fix-up pass over the 20 KiB
// it has just written (L2-hot).  Per-state reductions (edge lnL, derivatives)
// are xor-16/xor-32 shuffles over the four q-groups of a wave.
//
// Roofline: inner x inner site-update = 480 B + 12/R B and
// 1620 flops -> AI 3.4 flop/B: HBM-bound.  At 6.3 TB/s the MFMA pipe is ~42 %
// busy (40 MFMA per unit incl. the padded rows 20..31 of the second M-tile).
#pragma once

#include "kernels_common.hpp"
#include "kernels_generic.hpp"
#include "engine.h"
#include <cstdlib>

namespace pllhip {

constexpr unsigned S20_BS = 32;              // sites per block
constexpr unsigned S20_UNIT = 20 * S20_BS;   // doubles per (block, rate) unit
constexpr unsigned S20_FRAGS = 10 * 64;      // A-fragment doubles per (child, rate)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// streaming accesses: CLV data is touched once per operation, so it can bypass
// the cache allocation policy (flags bit 0: loads, bit 1: stores)
__device__ inline double2 s20_ld(const double * p, bool nt)
{
  const v2d v = nt ? __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p))
                   : *reinterpret_cast<const v2d *>(p);
  return make_double2(v.x, v.y);
}

__device__ inline void s20_st(double * p, const double2 & t, bool nt)
{
  v2d v;
  v.x = t.x;
  v.y = t.y;
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v2d *>(p));
  else *reinterpret_cast<v2d *>(p) = v;
}

__device__ inline v4d mfma_f64(double a, double b, v4d c)
{
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Rows 16 .. 19 of a 20-state product.  A second 16-row M tile would carry 12 rows of padding (and did until round 4:
// half of the matrix time of a unit).  v_mfma_f64_4x4x4_4b_f64 computes four independent 4 x 4 x 4 products, block
// b = (lane >> 2) & 3, with the k index in the lane's upper two bits -- exactly where this family keeps it: with
// lane = 16 q + n the instruction reads A_b[i][k] from lane (k = q, i = n & 3), B_b[k][j] from lane (k = q, column
// n = 4 b + j) and leaves D_b[i][j] in lane (i = q, column n).  So the B operand is the register the 16 x 16 x 4
// sequence uses (state row 4 ks + q, sites 2n / 2n + 1), the A operand is M[16 + (n & 3)][4 ks + q] for every block,
// and the result lands where the D layout wants it: row 16 + q, column n -- one double instead of a v4d of which one
// element was used.  A quarter of the matrix-pipe time of the padded tile; bit-identical results (the same four-term
// inner sum per k-step: tools/micro/mfma_4x4x4.hip compares the two instructions on random data).
__device__ inline double mfma_f64_tail(double a, double b, double c)
{
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// row (state) held by D slot k of lane group q: k = 0..3 -> first M-tile
// registers, k = 4 -> register 0 of the second M-tile (states 16..19)
__device__ inline unsigned s20_row(unsigned k, unsigned q) { return (k < 4) ? q + 4 * k : 16 + q; }

// A fragments of one 20x20 row-major matrix set [R][20][20] into LDS:
//   frag[((r*2 + mt)*5 + ks)*64 + lane] = M[r][ (lane&15)+16*mt ][ 4*ks + (lane>>4) ]  (0 beyond row 19)
__device__ inline void s20_fill_frags(double * frag, const double * mats, unsigned R)
{
  // (staged_loop, kernels_common.hpp: the loads of a thread go out together)
  staged_loop<10>(R * S20_FRAGS, [=](unsigned e)
  {
    const unsigned lane = e & 63, f = e >> 6, ks = f % 5, mt = (f / 5) & 1, r = f / 10;
    const unsigned i = (lane & 15) + 16 * mt, j = 4 * ks + (lane >> 4);
    const double x = mats[((size_t)r * 20 + (i < 20 ? i : 0)) * 20 + j];
    return (i < 20) ? x : 0.0;
  }, [=](unsigned e, double x) { frag[e] = x; });
}

// child term in D layout: t[k] = {even site, odd site} for row s20_row(k, q)
__device__ inline void s20_child_inner(const double * unit, const double * frag_r, unsigned lane,
                                       double2 t[5], bool nt = false)
{
  const unsigned off = lane * 2;            // doubles: (q*32 + 2n)
  double2 b[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
    b[ks] = s20_ld(unit + ks * 128 + off, nt);
  v4d a0e = {0, 0, 0, 0}, a0o = {0, 0, 0, 0};
  double a1e = 0.0, a1o = 0.0;
  // (rows 16 .. 19: the fragment of lane 16 q + (n & 3), see mfma_f64_tail)
  const double * tail_r = frag_r + 5 * 64 + (lane & ~12u);
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
  {
    const double f0 = frag_r[ks * 64 + lane];
    const double f1 = tail_r[ks * 64];
    a0e = mfma_f64(f0, b[ks].x, a0e);
    a0o = mfma_f64(f0, b[ks].y, a0o);
    a1e = mfma_f64_tail(f1, b[ks].x, a1e);
    a1o = mfma_f64_tail(f1, b[ks].y, a1o);
  }
  t[0] = make_double2(a0e[0], a0o[0]);
  t[1] = make_double2(a0e[1], a0o[1]);
  t[2] = make_double2(a0e[2], a0o[2]);
  t[3] = make_double2(a0e[3], a0o[3]);
  t[4] = make_double2(a1e, a1o);
}

// row stride of a lookup table staged in LDS: 21 doubles -- with 20, codes that differ by a
// multiple of 8 share all their banks (40 * code mod 64)
constexpr unsigned S20_LUT_RS = 21;

__device__ inline void s20_child_tip(const double * lut_r, unsigned code_e, unsigned code_o,
                                     unsigned q, double2 t[5], unsigned stride = 20)
{
  const double * le = lut_r + code_e * stride, * lo = lut_r + code_o * stride;
#pragma unroll
  for (unsigned k = 0; k < 5; ++k)
  {
    const unsigned i = s20_row(k, q);
    t[k] = make_double2(le[i], lo[i]);
  }
}

// load a unit in D layout (rows s20_row(k,q)) -- same addresses the stores use
__device__ inline void s20_load_d(const double * unit, unsigned lane, double2 t[5])
{
  const unsigned off = lane * 2;
#pragma unroll
  for (int k = 0; k < 5; ++k)
    t[k] = *reinterpret_cast<const double2 *>(unit + k * 128 + off);
}

__device__ inline void s20_store_d(double * unit, unsigned lane, const double2 t[5], bool nt = false)
{
  const unsigned off = lane * 2;
#pragma unroll
  for (int k = 0; k < 5; ++k)
    s20_st(unit + k * 128 + off, t[k], nt);
}

__device__ inline void s20_tip_d(unsigned long long mask_e, unsigned long long mask_o, unsigned q,
                                 double2 t[5])
{
#pragma unroll
  for (unsigned k = 0; k < 5; ++k)
  {
    const unsigned i = s20_row(k, q);
    t[k] = make_double2((double)((mask_e >> i) & 1ULL), (double)((mask_o >> i) & 1ULL));
  }
}

// AND over the four q-groups (lanes l, l^16, l^32, l^48)
__device__ inline int s20_and_q(int v)
{
  v &= __shfl_xor(v, 16, 64);
  v &= __shfl_xor(v, 32, 64);
  return v;
}

__device__ inline double s20_sum_q(double v)
{
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------------------
// partials (also used for the sumtable, with eigen-basis matrices in place of
// the P-matrices).   grid = (gx, ops), block = 256 (4 independent waves)
// dynamic LDS = 2 * R * S20_FRAGS doubles
// ---------------------------------------------------------------------------
template <unsigned RT>   // RT > 0: rate count known at compile time (register-resident block)
__device__ inline void s20_op_body(const OpDesc & op, unsigned nblk, unsigned Rrt,
                                   unsigned lut_codes, unsigned flags, double * frag)
{
  const bool nt_ld = flags & 1u, nt_st = flags & 2u;
  const bool rate_scalers = flags & 4u;      // PLL_ATTRIB_RATE_SCALERS: the vote covers one unit, scaler[n*R + r]
  const unsigned R = RT ? RT : Rrt;
  // each child owns R * S20_FRAGS doubles of LDS: the A fragments of its P-matrix,
  // or -- for a coded tip -- its lookup table (fits while lut_codes <= 30, rows padded to 21); LUT
  // gathers then hit LDS banks instead of the L1 address pipeline, which is what
  // bounds tip x tip operations otherwise
  const bool lut_lds = lut_codes * S20_LUT_RS <= S20_FRAGS;
  double * const frag2 = frag + R * S20_FRAGS;
  if (!op.codes1) s20_fill_frags(frag, op.pmat1, R);
  else if (lut_lds)
    staged_loop<8>(R * lut_codes * 20, [=](unsigned e) { return op.lut1[e]; }, [=](unsigned e, double x) { frag[(e / 20) * S20_LUT_RS + e % 20] = x; });
  if (!op.codes2) s20_fill_frags(frag2, op.pmat2, R);
  else if (lut_lds)
    staged_loop<8>(R * lut_codes * 20, [=](unsigned e) { return op.lut2[e]; }, [=](unsigned e, double x) { frag2[(e / 20) * S20_LUT_RS + e % 20] = x; });
  __syncthreads();

  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned wstride = gridDim.x * 4;
  const bool scaling = op.parent_scaler != nullptr;

  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += wstride)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    unsigned c1e = 0, c1o = 0, c2e = 0, c2o = 0;
    if (op.codes1) { c1e = op.codes1[site0]; c1o = op.codes1[site0 + 1]; }
    if (op.codes2) { c2e = op.codes2[site0]; c2o = op.codes2[site0 + 1]; }
    int small_e = 1, small_o = 1;

    if constexpr (RT > 0)
    {
      // compile-time rate count: all RT units of the block stay in registers
      // until the scaling vote is known, so nothing is written twice
      double2 out[RT][5];
#pragma unroll
      for (unsigned r = 0; r < RT; ++r)
      {
        const size_t ubase = ((size_t)blk * RT + r) * S20_UNIT;
        double2 t2[5];
        if (!op.codes1) s20_child_inner(op.clv1 + ubase, frag + r * S20_FRAGS, lane, out[r], nt_ld);
        else if (lut_lds) s20_child_tip(frag + r * lut_codes * S20_LUT_RS, c1e, c1o, q, out[r], S20_LUT_RS);
        else s20_child_tip(op.lut1 + (size_t)r * lut_codes * 20, c1e, c1o, q, out[r]);
        if (!op.codes2) s20_child_inner(op.clv2 + ubase, frag2 + r * S20_FRAGS, lane, t2, nt_ld);
        else if (lut_lds) s20_child_tip(frag2 + r * lut_codes * S20_LUT_RS, c2e, c2o, q, t2, S20_LUT_RS);
        else s20_child_tip(op.lut2 + (size_t)r * lut_codes * 20, c2e, c2o, q, t2);
        int re = 1, ro = 1;
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
          out[r][k].x *= t2[k].x;
          out[r][k].y *= t2[k].y;
          re &= (out[r][k].x < SCALE_THRESHOLD);
          ro &= (out[r][k].y < SCALE_THRESHOLD);
        }
        if (scaling && rate_scalers)
        {
          // per-rate scalers: decide, scale and count this unit now
          re = s20_and_q(re);
          ro = s20_and_q(ro);
          const double ue = re ? SCALE_FACTOR : 1.0, uo = ro ? SCALE_FACTOR : 1.0;
#pragma unroll
          for (int k = 0; k < 5; ++k) { out[r][k].x *= ue; out[r][k].y *= uo; }
          if (q == 0)
          {
            const size_t xe = site0 * RT + r, xo = (site0 + 1) * RT + r;
            unsigned ce = re ? 1u : 0u, co = ro ? 1u : 0u;
            if (op.scaler1) { ce += op.scaler1[xe]; co += op.scaler1[xo]; }
            if (op.scaler2) { ce += op.scaler2[xe]; co += op.scaler2[xo]; }
            op.parent_scaler[xe] = ce;
            op.parent_scaler[xo] = co;
          }
        }
        small_e &= re;
        small_o &= ro;
      }
      double fe = 1.0, fo = 1.0;
      if (scaling && !rate_scalers)
      {
        small_e = s20_and_q(small_e);
        small_o = s20_and_q(small_o);
        fe = small_e ? SCALE_FACTOR : 1.0;
        fo = small_o ? SCALE_FACTOR : 1.0;
      }
#pragma unroll
      for (unsigned r = 0; r < RT; ++r)
      {
#pragma unroll
        for (int k = 0; k < 5; ++k) { out[r][k].x *= fe; out[r][k].y *= fo; }
        s20_store_d(op.parent + ((size_t)blk * RT + r) * S20_UNIT, lane, out[r], nt_st);
      }
      if (scaling && !rate_scalers && q == 0)
      {
        unsigned ce = small_e ? 1u : 0u, co = small_o ? 1u : 0u;
        if (op.scaler1) { ce += op.scaler1[site0]; co += op.scaler1[site0 + 1]; }
        if (op.scaler2) { ce += op.scaler2[site0]; co += op.scaler2[site0 + 1]; }
        op.parent_scaler[site0] = ce;
        op.parent_scaler[site0 + 1] = co;
      }
      continue;
    }

    for (unsigned r = 0; r < R; ++r)
    {
      const size_t ubase = ((size_t)blk * R + r) * S20_UNIT;
      double2 t1[5], t2[5];
      if (!op.codes1) s20_child_inner(op.clv1 + ubase, frag + r * S20_FRAGS, lane, t1, nt_ld);
      else if (lut_lds) s20_child_tip(frag + r * lut_codes * S20_LUT_RS, c1e, c1o, q, t1, S20_LUT_RS);
      else s20_child_tip(op.lut1 + (size_t)r * lut_codes * 20, c1e, c1o, q, t1);
      if (!op.codes2) s20_child_inner(op.clv2 + ubase, frag2 + r * S20_FRAGS, lane, t2, nt_ld);
      else if (lut_lds) s20_child_tip(frag2 + r * lut_codes * S20_LUT_RS, c2e, c2o, q, t2, S20_LUT_RS);
      else s20_child_tip(op.lut2 + (size_t)r * lut_codes * 20, c2e, c2o, q, t2);
      int re = 1, ro = 1;
#pragma unroll
      for (int k = 0; k < 5; ++k)
      {
        t1[k].x *= t2[k].x;
        t1[k].y *= t2[k].y;
        re &= (t1[k].x < SCALE_THRESHOLD);
        ro &= (t1[k].y < SCALE_THRESHOLD);
      }
      if (scaling && rate_scalers)
      {
        re = s20_and_q(re);
        ro = s20_and_q(ro);
        const double ue = re ? SCALE_FACTOR : 1.0, uo = ro ? SCALE_FACTOR : 1.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) { t1[k].x *= ue; t1[k].y *= uo; }
        if (q == 0)
        {
          const size_t xe = site0 * R + r, xo = (site0 + 1) * R + r;
          unsigned ce = re ? 1u : 0u, co = ro ? 1u : 0u;
          if (op.scaler1) { ce += op.scaler1[xe]; co += op.scaler1[xo]; }
          if (op.scaler2) { ce += op.scaler2[xe]; co += op.scaler2[xo]; }
          op.parent_scaler[xe] = ce;
          op.parent_scaler[xo] = co;
        }
      }
      small_e &= re;
      small_o &= ro;
      s20_store_d(op.parent + ubase, lane, t1, nt_st);
    }

    if (scaling && !rate_scalers)
    {
      small_e = s20_and_q(small_e);
      small_o = s20_and_q(small_o);
      if (__any(small_e | small_o))
      {
        // rare: bring the just-written units of the flagged sites up by 2^256
        const double fe = small_e ? SCALE_FACTOR : 1.0, fo = small_o ? SCALE_FACTOR : 1.0;
        for (unsigned r = 0; r < R; ++r)
        {
          double * unit = op.parent + ((size_t)blk * R + r) * S20_UNIT;
          double2 t[5];
          s20_load_d(unit, lane, t);
#pragma unroll
          for (int k = 0; k < 5; ++k) { t[k].x *= fe; t[k].y *= fo; }
          s20_store_d(unit, lane, t);
        }
      }
      if (q == 0)
      {
        unsigned ce = small_e ? 1u : 0u, co = small_o ? 1u : 0u;
        if (op.scaler1) { ce += op.scaler1[site0]; co += op.scaler1[site0 + 1]; }
        if (op.scaler2) { ce += op.scaler2[site0]; co += op.scaler2[site0 + 1]; }
        op.parent_scaler[site0] = ce;
        op.parent_scaler[site0 + 1] = co;
      }
    }
  }
}

// one launch per dependency level: grid = (gx, ops of the level)
template <unsigned RT>
__global__ __launch_bounds__(256, 2) void k_partials_s20(OpBatch batch, unsigned nblk, unsigned Rrt,
                                                         unsigned lut_codes, unsigned flags)
{
  extern __shared__ double frag[];
  s20_op_body<RT>(batch.op[blockIdx.y], nblk, Rrt, lut_codes, flags, frag);
}

// ---------------------------------------------------------------------------
// Operation chains.  A chain is a run of operations in which each one consumes the
// parent vector of the one before (the path from a node towards the root).  Because
// the MFMA D layout of a result IS the B layout of an operand (slot k = state row
// s20_row(k, q) in both), a wave keeps the block it has just computed in registers
// and multiplies it straight into the next operation: the carried child is never
// re-read, which removes one of the three HBM streams of an inner x inner operation.
// Every vector is still stored (later partial traversals need it), bit-identical to
// what k_partials_s20 stores (k_chain_s20 below).
// ---------------------------------------------------------------------------

// LDS economy of the chain kernel.  The second M-tile of a 20 x 20 matrix holds rows 16..19
// only, i.e. just the lanes with (lane & 15) < 4 carry a value: those 16 lanes per k-step
// are stored densely (80 instead of 320 doubles per rate), the others use a literal 0.  A
// tip table is staged with the codes in use and rows of S20_LUT_RS doubles.  Per child:
//   inner: RT * S20_CFRAGS doubles (12.5 KiB for four rates instead of 20 KiB)
//   tip:   RT * codes * S20_LUT_RS doubles (13.8 KiB for 21 codes instead of 20 KiB)
// so that five to six operations fit the 160 KiB instead of four.
constexpr unsigned S20_CFRAGS = 5 * 64 + 5 * 16;      // compact A fragments per (child, rate)
constexpr unsigned S20_CHAIN_LDS = 20480;             // doubles: the whole LDS of a CU
constexpr unsigned S20_CHAIN_MAX = 8;                 // operations per chain (the LDS usually ends it earlier)
constexpr unsigned S20_CHAIN_WAVES = 8;

// Compact fragments for the 4 x 4 x 4 matrix instruction (mfma_f64_tail above): row group g = rows 4 g .. 4 g + 3,
// five groups for 20 states, no padding at all.  Lane (q, n) reads the A operand of (k-step ks, group g) at
//   cfrag[r*400 + (ks*5 + g)*16 + q*4 + (n & 3)] = M[r][4 g + (n & 3)][4 ks + q]
// (sixteen consecutive doubles per wave read: no bank conflicts, the four lanes of a (q, i) share an address).
__device__ inline void s20_cfrag_index(unsigned x, unsigned & row, unsigned & col)
{
  const unsigned t = x >> 4, ks = t / 5, g = t - 5 * ks;
  row = 4 * g + (x & 3u);
  col = 4 * ks + ((x >> 2) & 3u);
}

__device__ inline void s20_fill_cfrags(double * cfrag, const double * mats, unsigned R)
{
  staged_loop<8>(R * S20_CFRAGS, [=](unsigned e)
  {
    const unsigned r = e / S20_CFRAGS, x = e % S20_CFRAGS;
    unsigned i, j;
    s20_cfrag_index(x, i, j);
    return mats[((size_t)r * 20 + i) * 20 + j];
  }, [=](unsigned e, double x) { cfrag[e] = x; });
}

// child term from a B operand in registers (b[ks] = rows 4 ks + q of the child vector)
__device__ inline void s20_child_regs_c(const double2 b[5], const double * cfrag_r, unsigned lane,
                                        double2 t[5])
{
  // rows 0 .. 15: one 16 x 16 x 4 tile (its A operand M[n][4 ks + q] sits at group n >> 2, row n & 3 of the compact
  // fragments); rows 16 .. 19: the 4 x 4 x 4 instruction (mfma_f64_tail).  All five groups on the short instruction
  // were measured too: the same at 1 M sites (29.8 against 30.0 ms), 3.5 % slower at the 125 k-site slice (25 instead
  // of 10 LDS operand reads per unit).
  const unsigned q = lane >> 4, n = lane & 15;
  const double * head_r = cfrag_r + (n >> 2) * 16 + q * 4 + (n & 3u);
  const double * tail_r = cfrag_r + 4 * 16 + q * 4 + (n & 3u);
  v4d a0e = {0, 0, 0, 0}, a0o = {0, 0, 0, 0};
  double a1e = 0.0, a1o = 0.0;
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
  {
    const double f0 = head_r[ks * 80];
    const double f1 = tail_r[ks * 80];
    a0e = mfma_f64(f0, b[ks].x, a0e);
    a0o = mfma_f64(f0, b[ks].y, a0o);
    a1e = mfma_f64_tail(f1, b[ks].x, a1e);
    a1o = mfma_f64_tail(f1, b[ks].y, a1o);
  }
  t[0] = make_double2(a0e[0], a0o[0]);
  t[1] = make_double2(a0e[1], a0o[1]);
  t[2] = make_double2(a0e[2], a0o[2]);
  t[3] = make_double2(a0e[3], a0o[3]);
  t[4] = make_double2(a1e, a1o);
}

__device__ inline void s20_child_inner_c(const double * unit, const double * cfrag_r, unsigned lane,
                                         double2 t[5], bool nt)
{
  double2 b[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
    b[ks] = s20_ld(unit + ks * 128 + lane * 2, nt);
  s20_child_regs_c(b, cfrag_r, lane, t);
}

__device__ inline const unsigned * s20_wide_codes(const double * clv, const uint8_t * codes, const double * pfrag)
{
  return (!clv && !codes) ? reinterpret_cast<const unsigned *>(pfrag) : nullptr;
}

// one operation for one site block; X holds the handed-over operand on entry (when
// carried != 0) and the result on exit.  s1 / s2: the children's LDS tables.
// xe / xo: the scaler counts that go with X (per rate with RS = PLL_ATTRIB_RATE_SCALERS, where
// the vote covers one (site, rate) unit and the counts live at scaler[site * R + rate];
// otherwise one count per site, held in element 0).
// WIDE: the schedule may hold wide tips (site repeats); without them the instantiation carries none of their
// pointers and branches (the attribute-off kernel is the round-2 one: 13 instead of 19 spilled registers).
// store: false = the result is handed to the next operation of the chain in registers only (an evaluate-only
// traversal, PlanOp::flags bit 0); its scaler counts are written either way.
template <unsigned RT, bool RS, bool WIDE>
__device__ inline void s20_chain_op(const OpDesc & op, unsigned carried, double2 X[RT][5],
                                    const double * s1, const double * s2,
                                    unsigned lut_codes, unsigned lut_used, bool lut_lds,
                                    unsigned blk, unsigned lane, bool nt_ld, bool nt_st,
                                    unsigned (&xe)[RS ? RT : 1], unsigned (&xo)[RS ? RT : 1], bool store = true,
                                    unsigned wide_lds = 0)
{
  // wide_lds: PlanOp::flags -- bit 1 / 2: the rows of wide tip 1 / 2 are staged in LDS (s1 / s2, rows of S20_LUT_RS)
  const unsigned q = lane >> 4, n = lane & 15;
  const size_t site0 = (size_t)blk * S20_BS + 2 * n;
  unsigned c1e = 0, c1o = 0, c2e = 0, c2o = 0;
  // a "wide tip" (kernels_repeats.hpp: a cherry known per class of sites): neither vector nor byte codes; the
  // pfrag field holds its class codes (32 bits each), lut its table, childN_index the rows of that table
  const unsigned * w1 = WIDE ? s20_wide_codes(op.clv1, op.codes1, op.pfrag1) : nullptr;
  const unsigned * w2 = WIDE ? s20_wide_codes(op.clv2, op.codes2, op.pfrag2) : nullptr;
  if (op.codes1) { c1e = op.codes1[site0]; c1o = op.codes1[site0 + 1]; }
  else if (w1) { c1e = w1[site0]; c1o = w1[site0 + 1]; }
  if (op.codes2) { c2e = op.codes2[site0]; c2o = op.codes2[site0 + 1]; }
  else if (w2) { c2e = w2[site0]; c2o = w2[site0 + 1]; }
  const bool scaling = op.parent_scaler != nullptr;
  int small_e = 1, small_o = 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    const size_t ubase = ((size_t)blk * RT + r) * S20_UNIT;
    double2 t1[5], t2[5];
    if (carried == 1) s20_child_regs_c(X[r], s1 + r * S20_CFRAGS, lane, t1);
    else if (w1 && (wide_lds & 2u)) s20_child_tip(s1 + r * op.child1_index * S20_LUT_RS, c1e, c1o, q, t1, S20_LUT_RS);
    else if (w1) s20_child_tip(op.lut1 + (size_t)r * op.child1_index * 20, c1e, c1o, q, t1);
    else if (!op.codes1) s20_child_inner_c(op.clv1 + ubase, s1 + r * S20_CFRAGS, lane, t1, nt_ld);
    else if (lut_lds) s20_child_tip(s1 + r * lut_used * S20_LUT_RS, c1e, c1o, q, t1, S20_LUT_RS);
    else s20_child_tip(op.lut1 + (size_t)r * lut_codes * 20, c1e, c1o, q, t1);
    if (carried == 2) s20_child_regs_c(X[r], s2 + r * S20_CFRAGS, lane, t2);
    else if (w2 && (wide_lds & 4u)) s20_child_tip(s2 + r * op.child2_index * S20_LUT_RS, c2e, c2o, q, t2, S20_LUT_RS);
    else if (w2) s20_child_tip(op.lut2 + (size_t)r * op.child2_index * 20, c2e, c2o, q, t2);
    else if (!op.codes2) s20_child_inner_c(op.clv2 + ubase, s2 + r * S20_CFRAGS, lane, t2, nt_ld);
    else if (lut_lds) s20_child_tip(s2 + r * lut_used * S20_LUT_RS, c2e, c2o, q, t2, S20_LUT_RS);
    else s20_child_tip(op.lut2 + (size_t)r * lut_codes * 20, c2e, c2o, q, t2);
    // X[r] has been consumed (if it was an operand at all): it now takes the result
    int re = 1, ro = 1;
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      X[r][k].x = t1[k].x * t2[k].x;
      X[r][k].y = t1[k].y * t2[k].y;
      re &= (X[r][k].x < SCALE_THRESHOLD);
      ro &= (X[r][k].y < SCALE_THRESHOLD);
    }
    if (!RS)
    {
      small_e &= re;
      small_o &= ro;
      continue;
    }
    // per-rate scalers: the vote covers this unit alone -- decide, scale, store, count
    unsigned ce = 0, co = 0;
    if (scaling)
    {
      const int se = s20_and_q(re), so = s20_and_q(ro);
      const double fe = se ? SCALE_FACTOR : 1.0, fo = so ? SCALE_FACTOR : 1.0;
#pragma unroll
      for (int k = 0; k < 5; ++k)
      {
        X[r][k].x *= fe;
        X[r][k].y *= fo;
      }
      if (q == 0)
      {
        const size_t ie = site0 * RT + r, io = (site0 + 1) * RT + r;
        ce = se ? 1u : 0u;
        co = so ? 1u : 0u;
        if (op.scaler1)
        {
          if (carried == 1) { ce += xe[RS ? r : 0]; co += xo[RS ? r : 0]; }
          else { ce += op.scaler1[ie]; co += op.scaler1[io]; }
        }
        if (op.scaler2)
        {
          if (carried == 2) { ce += xe[RS ? r : 0]; co += xo[RS ? r : 0]; }
          else { ce += op.scaler2[ie]; co += op.scaler2[io]; }
        }
        op.parent_scaler[ie] = ce;
        op.parent_scaler[io] = co;
      }
    }
    if (store) s20_store_d(op.parent + ubase, lane, X[r], nt_st);
    xe[RS ? r : 0] = ce;
    xo[RS ? r : 0] = co;
  }
  if (RS) return;
  double fe = 1.0, fo = 1.0;
  if (scaling)
  {
    small_e = s20_and_q(small_e);
    small_o = s20_and_q(small_o);
    fe = small_e ? SCALE_FACTOR : 1.0;
    fo = small_o ? SCALE_FACTOR : 1.0;
  }
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      X[r][k].x *= fe;
      X[r][k].y *= fo;
    }
    if (store) s20_store_d(op.parent + ((size_t)blk * RT + r) * S20_UNIT, lane, X[r], nt_st);
  }
  unsigned ce = 0, co = 0;
  if (scaling && q == 0)
  {
    ce = small_e ? 1u : 0u;
    co = small_o ? 1u : 0u;
    // (a wide tip's counts are kept per class, like its vector: indexed by the class codes)
    if (op.scaler1)
    {
      if (carried == 1) { ce += xe[0]; co += xo[0]; }
      else if (w1) { ce += op.scaler1[c1e]; co += op.scaler1[c1o]; }
      else { ce += op.scaler1[site0]; co += op.scaler1[site0 + 1]; }
    }
    if (op.scaler2)
    {
      if (carried == 2) { ce += xe[0]; co += xo[0]; }
      else if (w2) { ce += op.scaler2[c2e]; co += op.scaler2[c2o]; }
      else { ce += op.scaler2[site0]; co += op.scaler2[site0 + 1]; }
    }
    op.parent_scaler[site0] = ce;
    op.parent_scaler[site0 + 1] = co;
  }
  xe[0] = ce;
  xo[0] = co;
}

// row-major [R][20][20] matrix sets -> compact fragment order (after a host upload of the
// P-matrices; k_pmatrix writes both forms itself).  grid = matrices, block = 256
__global__ __launch_bounds__(256) void k_s20_pfrag(const double * pmat, double * pfrag, unsigned R)
{
  s20_fill_cfrags(pfrag + (size_t)blockIdx.x * R * S20_CFRAGS, pmat + (size_t)blockIdx.x * R * 400, R);
}

// stage one child's table: compact fragments, or the rows of the codes in use of a tip table
__device__ inline void s20_fill_slot(double * slot, const double * pmat, const double * pfrag, const double * lut,
                                     unsigned R, unsigned lut_codes, unsigned lut_used, bool lut_lds)
{
  if (!lut)
  {
    if (pfrag)        // written in fragment order by k_pmatrix: a plain, coalesced copy
      staged_loop<4>(R * S20_CFRAGS / 2, [=](unsigned e) { return reinterpret_cast<const double2 *>(pfrag)[e]; },
                     [=](unsigned e, double2 v) { reinterpret_cast<double2 *>(slot)[e] = v; });
    else s20_fill_cfrags(slot, pmat, R);
  }
  else if (lut_lds)
  {
    // (index arithmetic with constant divisors only: a division by the run-time row count per element
    // made this loop a third of a chain's staging time)
    const unsigned per_rate = lut_used * 20;
    for (unsigned x = threadIdx.x; x < per_rate; x += blockDim.x)
    {
      const unsigned c = x / 20, i = x - c * 20;
      for (unsigned r0 = 0; r0 < R; r0 += 4)          // the loads of (up to) four rates go out together
      {
        double v[4];
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) v[u] = lut[(size_t)(r0 + u < R ? r0 + u : r0) * lut_codes * 20 + x];
#pragma unroll
        for (unsigned u = 0; u < 4; ++u)
          if (r0 + u < R) slot[((r0 + u) * lut_used + c) * S20_LUT_RS + i] = v[u];
      }
    }
  }
}

// Operation chains: the tables of ALL operations of a chain stay in LDS, so -- exactly like
// k_partials_s20 -- a workgroup fills LDS once and its waves then stream site blocks
// independently, without any barrier; per block a wave runs the chain bottom-up with the
// intermediate vector in registers.  One 512-thread workgroup per CU (eight waves share the
// tables: same occupancy as two 256-thread workgroups, half the LDS).
// grid = (gx, chains), block = 512, dynamic LDS = the largest chain area of the launch.
template <unsigned RT, bool RS>
__global__ __launch_bounds__(64 * S20_CHAIN_WAVES, 1) void k_chain_s20(ChainBatch batch, unsigned nblk,
                                                                        unsigned lut_codes, unsigned lut_used,
                                                                        unsigned flags)
{
  extern __shared__ double lds[];
  const bool nt_ld = flags & 1u, nt_st = flags & 2u;
  const bool lut_lds = (flags & 8u) != 0;             // tip tables are staged (decided by the host)
  const unsigned first = batch.first[blockIdx.y], len = batch.len[blockIdx.y];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned i = 0; i < len; ++i)
  {
    const OpDesc & op = batch.op[first + i];
    s20_fill_slot(lds + batch.slot1[first + i], op.pmat1, op.pfrag1, op.codes1 ? op.lut1 : nullptr, RT, lut_codes, lut_used, lut_lds);
    s20_fill_slot(lds + batch.slot2[first + i], op.pmat2, op.pfrag2, op.codes2 ? op.lut2 : nullptr, RT, lut_codes, lut_used, lut_lds);
  }
  __syncthreads();

  const unsigned wstride = gridDim.x * S20_CHAIN_WAVES;
  for (unsigned blk = blockIdx.x * S20_CHAIN_WAVES + wave; blk < nblk; blk += wstride)
  {
    double2 X[RT][5];
    unsigned xe[RS ? RT : 1] = {}, xo[RS ? RT : 1] = {};
#pragma unroll 1
    for (unsigned i = 0; i < len; ++i)
      s20_chain_op<RT, RS, false>(batch.op[first + i], i ? batch.carried[first + i] : 0u, X,
                       lds + batch.slot1[first + i], lds + batch.slot2[first + i],
                       lut_codes, lut_used, lut_lds, blk, lane, nt_ld, nt_st, xe, xo);
  }
}

// A whole traversal in one launch (PlanView, engine.h): the workgroup walks every chain of the
// schedule in dependency order.  A wave keeps the SAME site blocks in every chain, so all it
// reads from an earlier chain it has written itself -- there is nothing to wait for but the
// workgroup's own barrier around the re-staging of the LDS tables.
// grid = gx, block = 512, dynamic LDS = the largest chain area of the schedule.
// TRANS: an evaluate-only traversal -- operations whose PlanOp::flags bit 0 is set hand their result on in registers only
// (a compile-time switch: as a run-time flag in front of every store it cost the storing traversal 3.5 %)
template <unsigned RT, bool RS, bool WIDE, bool TRANS>
__global__ __launch_bounds__(64 * S20_CHAIN_WAVES, 1) void k_traverse_s20(PlanView plan, unsigned chain_begin,
                                                                           unsigned chain_end, unsigned nblk,
                                                                           unsigned slab, unsigned flags)
{
  extern __shared__ double lds[];
  const bool nt_ld = flags & 1u, nt_st = flags & 2u;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wstride = gridDim.x * S20_CHAIN_WAVES;
  const PlanOp * plan_ops_;
  const PlanChain * plan_chains_;
  plan_bases(plan, plan_ops_, plan_chains_);
  bool first_fill = true;
  // slabs of `slab` site blocks go through the whole schedule one after the other
  // (nblk: the largest partition of the schedule; a chain knows its own partition's extent and tables)
  for (unsigned s0 = 0; s0 < nblk; s0 += slab)
  {
    // chains [chain_begin, chain_end) of the schedule, shared out over gridDim.y: the whole schedule with
    // gridDim.y = 1 (a traversal in one launch), one chain per workgroup row for a round of chains
    for (unsigned c = chain_begin + blockIdx.y; c < chain_end; c += gridDim.y)
    {
      const PlanChain ch = plan_fetch(plan_chains_ + c);
      const unsigned s1 = min(ch.extent, s0 + slab);
      const unsigned lut_codes = ch.lut_codes, lut_used = ch.lut_used;
      const bool lut_lds = (ch.flags & 1u) != 0;
      if (!first_fill) __syncthreads();                 // every wave has left the previous chain's tables
      first_fill = false;
      for (unsigned i = 0; i < ch.len; ++i)
      {
        const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
        // (a wide tip has its rows in LDS when they are few -- PlanOp::flags bit 1 / 2 --, else no table there)
        if (!WIDE || po.d.clv1 || po.d.codes1)
          s20_fill_slot(lds + po.slot1, po.d.pmat1, po.d.pfrag1, po.d.codes1 ? po.d.lut1 : nullptr, RT, lut_codes, lut_used, lut_lds);
        else if (po.flags & 2u)
          s20_fill_slot(lds + po.slot1, nullptr, nullptr, po.d.lut1, RT, po.d.child1_index, po.d.child1_index, true);
        if (!WIDE || po.d.clv2 || po.d.codes2)
          s20_fill_slot(lds + po.slot2, po.d.pmat2, po.d.pfrag2, po.d.codes2 ? po.d.lut2 : nullptr, RT, lut_codes, lut_used, lut_lds);
        else if (po.flags & 4u)
          s20_fill_slot(lds + po.slot2, nullptr, nullptr, po.d.lut2, RT, po.d.child2_index, po.d.child2_index, true);
      }
      __syncthreads();

      for (unsigned blk = s0 + blockIdx.x * S20_CHAIN_WAVES + wave; blk < s1; blk += wstride)
      {
        double2 X[RT][5];
        unsigned xe[RS ? RT : 1] = {}, xo[RS ? RT : 1] = {};
#pragma unroll 1
        for (unsigned i = 0; i < ch.len; ++i)
        {
          const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
          s20_chain_op<RT, RS, WIDE>(po.d, i ? po.carried : 0u, X, lds + po.slot1, lds + po.slot2,
                                     lut_codes, lut_used, lut_lds, blk, lane, nt_ld, nt_st, xe, xo, TRANS ? !(po.flags & 1u) : true,
                                     WIDE ? po.flags : 0u);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// edge / root log-likelihood.  grid = nblocks (<= REDUCE_BLOCKS), block = 256
// dynamic LDS = R * S20_FRAGS doubles (unused for the root form)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_edge_lnl_s20(ModelView mv, ParamIdx fidx,
                                                         NodeRef parent, NodeRef child,
                                                         const double * pmat, const double * lut,
                                                         unsigned lut_codes,
                                                         const unsigned * ps, const unsigned * cs,
                                                         const unsigned * weights, const int * invariant,
                                                         const unsigned long long * tipmap,
                                                         unsigned N, unsigned nblk, unsigned R,
                                                         double * persite, ReduceOut block_out,
                                                         unsigned rate_scalers)
{
  extern __shared__ double frag[];
  __shared__ double scratch[4];
  if (pmat && !child.codes) s20_fill_frags(frag, pmat, R);
  __syncthreads();

  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned wstride = gridDim.x * 4;
  double acc = 0.0;

  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += wstride)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    SiteSide sd = load_site_side(rate_scalers ? nullptr : ps, rate_scalers ? nullptr : cs, weights, site0, N,
                                 q == 0 && blk < nblk);
    if (rate_scalers)
    {
      sd.cnt_e = rate_min_count(ps, cs, site0, R);
      sd.cnt_o = rate_min_count(ps, cs, site0 + 1, R);
    }
    unsigned cce = 0, cco = 0;
    unsigned long long pme = 0, pmo = 0;
    if (child.codes) { cce = child.codes[site0]; cco = child.codes[site0 + 1]; }
    if (parent.codes) { pme = tipmap[parent.codes[site0]]; pmo = tipmap[parent.codes[site0 + 1]]; }
    double site_e = 0.0, site_o = 0.0, inv_e = 0.0, inv_o = 0.0;
    int inv_state_e = -1, inv_state_o = -1;
    if (invariant)
    {
      inv_state_e = (site0 < N) ? invariant[site0] : -1;
      inv_state_o = (site0 + 1 < N) ? invariant[site0 + 1] : -1;
    }

    for (unsigned r = 0; r < R; ++r)
    {
      const size_t ubase = ((size_t)blk * R + r) * S20_UNIT;
      const unsigned fi = fidx.v[r];
      const double * pi = mv.freqs(fi);
      double2 t[5], pv[5];
      if (!pmat)
      {
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] = make_double2(1.0, 1.0);
      }
      else if (child.codes) s20_child_tip(lut + (size_t)r * lut_codes * 20, cce, cco, q, t);
      else s20_child_inner(child.clv + ubase, frag + r * S20_FRAGS, lane, t);
      if (parent.codes) s20_tip_d(pme, pmo, q, pv);
      else s20_load_d(parent.clv + ubase, lane, pv);
      double le = 0.0, lo = 0.0;
#pragma unroll
      for (unsigned k = 0; k < 5; ++k)
      {
        const double f = pi[s20_row(k, q)];
        le += f * pv[k].x * t[k].x;
        lo += f * pv[k].y * t[k].y;
      }
      le = s20_sum_q(le);
      lo = s20_sum_q(lo);
      if (rate_scalers)
      {
        le *= rate_factor(ps, cs, site0, R, r, sd.cnt_e);
        lo *= rate_factor(ps, cs, site0 + 1, R, r, sd.cnt_o);
      }
      const double pinv = mv.pinv()[fi], w = mv.weights()[r];
      if (pinv > 0.0)
      {
        site_e += w * (1.0 - pinv) * le;
        site_o += w * (1.0 - pinv) * lo;
        if (inv_state_e >= 0) inv_e += w * pinv * pi[inv_state_e];
        if (inv_state_o >= 0) inv_o += w * pinv * pi[inv_state_o];
      }
      else
      {
        site_e += w * le;
        site_o += w * lo;
      }
    }
    if (q == 0)
    {
      if (site0 < N)
      {
        const unsigned cnt = sd.cnt_e;
        const double l = site_loglh(site_e, cnt, inv_e);
        if (persite) persite[site0] = l;
        acc += l * (double)sd.w_e;
      }
      if (site0 + 1 < N)
      {
        const unsigned cnt = sd.cnt_o;
        const double l = site_loglh(site_o, cnt, inv_o);
        if (persite) persite[site0 + 1] = l;
        acc += l * (double)sd.w_o;
      }
    }
  }
  const double tot = block_sum_256(acc, scratch);
  grid_reduce_finish1(tot, block_out, scratch);
}

// ---------------------------------------------------------------------------
// sumtable preparation: eigen-basis matrices in the [r][row][col] form the
// partials kernel consumes, plus their tip lookup tables
//   Lm[r][k][i] = pi_i V[i][k],   Rm[r][k][j] = V^-1[k][j]
//   lutL[r][code][k] = sum_{i in mask} Lm[r][k][i]   (same for R)
// grid = R, block = 256
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumtable_prep_s20(ModelView mv, ParamIdx params,
                                                           const unsigned long long * tipmap,
                                                           unsigned lut_codes, bool want_lut,
                                                           double * Lm, double * Rm,
                                                           double * lutL, double * lutR)
{
  const unsigned r = blockIdx.x, pi_ = params.v[r];
  const double * pi = mv.freqs(pi_), * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_);
  double * L = Lm + (size_t)r * 400, * Rr = Rm + (size_t)r * 400;
  for (unsigned e = threadIdx.x; e < 400; e += blockDim.x)
  {
    const unsigned k = e / 20, i = e % 20;
    L[e] = pi[i] * V[i * 20 + k];
    Rr[e] = Vi[k * 20 + i];
  }
  if (!want_lut) return;
  __syncthreads();
  for (unsigned e = threadIdx.x; e < lut_codes * 20; e += blockDim.x)
  {
    const unsigned c = e / 20, k = e % 20;
    const unsigned long long mask = tipmap[c];
    double a = 0.0, b = 0.0;
    if (mask && !(mask & (mask - 1)))                 // one state (most codes): no walk over the alphabet
    {
      const unsigned i = (unsigned)__ffsll((long long)mask) - 1;
      a += pi[i] * V[i * 20 + k];
      b += Vi[k * 20 + i];
    }
    else
      for (unsigned i = 0; i < 20; ++i)
        if ((mask >> i) & 1ULL) { a += pi[i] * V[i * 20 + k]; b += Vi[k * 20 + i]; }
    lutL[((size_t)r * lut_codes + c) * 20 + k] = a;
    lutR[((size_t)r * lut_codes + c) * 20 + k] = b;
  }
}

// ---------------------------------------------------------------------------
// derivatives of -lnL at up to FOUR trial branch lengths from ONE pass over a blocked
// sumtable, on the matrix cores (20- and 61-state families: KS = 5 / 16 k-steps).
//   A_n(t) = sum_r sum_k sum[n,r,k] e0_t[r][k],  B_n, C_n with e1, e2    
// is a [16 x rows] x [rows x 32 sites] product per rate, accumulated over the rates in one
// accumulator: row j + 4c of the left operand holds e_c of trial length j (c = 0, 1, 2;
// rows 12..15 are zero), and a unit of the sumtable IS the right operand (B layout of the
// family).  In the D layout lane group q then holds A, B, C (registers 0, 1, 2) of trial
// length j = q for its two sites: the per-site ratios need no cross-lane traffic at all.
// Four lengths cost what one costs (the kernel stays HBM-bound: 2 KS MFMAs per 1 KiB x KS
// loaded), and a row's result does not depend on the other rows, i.e. on which other
// lengths share the launch.
// dynamic LDS = R * KS * 64 doubles (the left operands as per-lane fragments)
// totals: df[0], ddf[0], df[1], ddf[1], ...
// ---------------------------------------------------------------------------
// the scan of one launch: block totals of {df, ddf} of trial length j in threads 2 j, 2 j + 1 (< 8) of the block
template <unsigned KS, unsigned SREAL>
__device__ inline double deriv_block_totals(const ModelView & mv, const ParamIdx & params, const TrialLengths & tl,
                                            unsigned ntrial, const double * sumtable,
                                            const unsigned * ps, const unsigned * cs,
                                            const unsigned * weights, const int * invariant,
                                            unsigned N, unsigned nblk, unsigned R, unsigned rate_scalers,
                                            double * frag, GridView gv = launch_grid())
{
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const unsigned SR = SREAL ? SREAL : mv.S;       // SREAL = 0: the state count is a run-time value
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned wstride = gv.G * 4;
  unsigned blk = gv.b * 4 + wave;
  double2 bufA[KS], bufB[KS];
  auto load_unit = [&](double2 (&b)[KS], unsigned blk_, unsigned r_)
  {
    const double * unit = sumtable + ((size_t)blk_ * R + r_) * UNIT + lane * 2;
#pragma unroll
    for (unsigned ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const double2 *>(unit + ks * 128);
  };
  if (blk < nblk) load_unit(bufA, blk, 0);
  // left operands: one exp per (rate, eigenvalue, trial length) and block -- the three powers
  // of lambda share it, rows of unused trial slots repeat the last length
  for (unsigned x = threadIdx.x; x < R * KS * 64; x += blockDim.x) frag[x] = 0.0;
  __syncthreads();
  for (unsigned x = threadIdx.x; x < R * SR * ntrial; x += blockDim.x)
  {
    const unsigned j = x % ntrial, k = (x / ntrial) % SR, r = x / (ntrial * SR);
    const unsigned pi_ = params.v[r];
    const double pinv = mv.pinv()[pi_];
    const double lam = mv.evals(pi_)[k] * mv.rates()[r] / (1.0 - pinv);
    const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
    const double ex = exp(lam * tl.t[j]);
    const double e0 = wr * ex, e1 = wr * ex * lam, e2 = wr * ex * lam * lam;
    double * f = frag + ((size_t)r * KS + (k >> 2)) * 64 + (k & 3) * 16;     // lane = (k & 3) * 16 + row
    for (unsigned jj = j; jj < 4; jj += (j + 1 == ntrial) ? 1u : 4u)
    {
      f[jj] = e0;
      f[jj + 4] = e1;
      f[jj + 8] = e2;
    }
  }
  __syncthreads();

  double df = 0.0, ddf = 0.0;                 // of trial length q
  // The wave walks its (site block, rate) units with two alternating operand buffers: the
  // loads of unit u + 1 are in flight while the MFMAs of unit u run (the very first unit was
  // requested before the tables above were built).
  double acc_e[3] = {0.0, 0.0, 0.0}, acc_o[3] = {0.0, 0.0, 0.0};       // {A, B, C} of trial length q
  double inv_e = 0, inv_o = 0;
  SiteSide sd = {0u, 0u, 0u, 0u};
  unsigned r = 0;
#define PLLHIP_DERIV_UNIT(CUR, NXT)                                                                   \
  {                                                                                                   \
    unsigned nr = r + 1, nb = blk;                                                                    \
    if (nr == R) { nr = 0; nb = blk + wstride; }                                                      \
    const bool more = nb < nblk;                                                                      \
    if (more) load_unit(NXT, nb, nr);                                                                 \
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;                                                \
    if (r == 0)                                                                                       \
    {                                                                                                 \
      sd = load_site_side(rate_scalers ? nullptr : ps, rate_scalers ? nullptr : cs, weights, site0, N, true); \
      if (rate_scalers)                                                                               \
      {                                                                                               \
        sd.cnt_e = rate_min_count(ps, cs, site0, R);                                                  \
        sd.cnt_o = rate_min_count(ps, cs, site0 + 1, R);                                              \
      }                                                                                               \
    }                                                                                                 \
    if (rate_scalers)                                                                                 \
    {                                                                                                 \
      /* per-rate counts: the unit is brought to the smallest count of its site first */             \
      const double fe = rate_factor(ps, cs, site0, R, r, sd.cnt_e);                                   \
      const double fo = rate_factor(ps, cs, site0 + 1, R, r, sd.cnt_o);                               \
      _Pragma("unroll") for (unsigned ks = 0; ks < KS; ++ks) { CUR[ks].x *= fe; CUR[ks].y *= fo; }    \
    }                                                                                                 \
    /* rows j, 4 + j, 8 + j of the left operand (e0, e1, e2 of trial length j) as three row groups of the  */ \
    /* 4 x 4 x 4 instruction (mfma_f64_tail): lane (q, n) reads row 4 g + (n & 3), k = 4 ks + q, and gets */ \
    /* quantity g of trial length q -- six independent short chains instead of two long ones            */ \
    const double * fr = frag + (size_t)r * KS * 64 + (lane >> 4) * 16 + (lane & 3u);                  \
    _Pragma("unroll") for (unsigned ks = 0; ks < KS; ++ks)                                            \
    {                                                                                                 \
      const double f0 = fr[ks * 64], f1 = fr[ks * 64 + 4], f2 = fr[ks * 64 + 8];                      \
      acc_e[0] = mfma_f64_tail(f0, CUR[ks].x, acc_e[0]);                                              \
      acc_o[0] = mfma_f64_tail(f0, CUR[ks].y, acc_o[0]);                                              \
      acc_e[1] = mfma_f64_tail(f1, CUR[ks].x, acc_e[1]);                                              \
      acc_o[1] = mfma_f64_tail(f1, CUR[ks].y, acc_o[1]);                                              \
      acc_e[2] = mfma_f64_tail(f2, CUR[ks].x, acc_e[2]);                                              \
      acc_o[2] = mfma_f64_tail(f2, CUR[ks].y, acc_o[2]);                                              \
    }                                                                                                 \
    if (invariant)                                                                                    \
    {                                                                                                 \
      const unsigned pi_ = params.v[r];                                                               \
      const double pinv = mv.pinv()[pi_];                                                             \
      if (pinv > 0.0)                                                                                 \
      {                                                                                               \
        const double w = mv.weights()[r] * pinv;                                                      \
        if (site0 < N && invariant[site0] >= 0) inv_e += w * mv.freqs(pi_)[invariant[site0]];        \
        if (site0 + 1 < N && invariant[site0 + 1] >= 0) inv_o += w * mv.freqs(pi_)[invariant[site0 + 1]]; \
      }                                                                                               \
    }                                                                                                 \
    if (r == R - 1)                                                                                   \
    {                                                                                                 \
      if (site0 < N)                                                                                  \
      {                                                                                               \
        double a = acc_e[0];                                                                          \
        if (inv_e > 0.0) a += (sd.cnt_e <= 3) ? ldexp(inv_e, 256 * (int)sd.cnt_e) : INFINITY;         \
        const double w = (double)sd.w_e, ba = acc_e[1] / a, ca = acc_e[2] / a;                        \
        df -= w * ba;                                                                                 \
        ddf += w * (ba * ba - ca);                                                                    \
      }                                                                                               \
      if (site0 + 1 < N)                                                                              \
      {                                                                                               \
        double a = acc_o[0];                                                                          \
        if (inv_o > 0.0) a += (sd.cnt_o <= 3) ? ldexp(inv_o, 256 * (int)sd.cnt_o) : INFINITY;         \
        const double w = (double)sd.w_o, ba = acc_o[1] / a, ca = acc_o[2] / a;                        \
        df -= w * ba;                                                                                 \
        ddf += w * (ba * ba - ca);                                                                    \
      }                                                                                               \
      acc_e[0] = acc_e[1] = acc_e[2] = 0.0;                                                           \
      acc_o[0] = acc_o[1] = acc_o[2] = 0.0;                                                           \
      inv_e = inv_o = 0.0;                                                                            \
    }                                                                                                 \
    blk = nb;                                                                                         \
    r = nr;                                                                                           \
    if (!more) break;                                                                                 \
  }
  while (blk < nblk)
  {
    PLLHIP_DERIV_UNIT(bufA, bufB)
    PLLHIP_DERIV_UNIT(bufB, bufA)
  }
#undef PLLHIP_DERIV_UNIT
  // lane group q holds trial length q: sums over the 16 lanes of a group (fixed butterfly), then
  // over the four waves through LDS -- one barrier pair whatever the number of lengths
  __shared__ double part[4][8];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1)
  {
    df += __shfl_xor(df, off, 64);
    ddf += __shfl_xor(ddf, off, 64);
  }
  if (n == 0) { part[wave][2 * q] = df; part[wave][2 * q + 1] = ddf; }
  __syncthreads();
  double mine = 0.0;
  if (threadIdx.x < 8) mine = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
  __syncthreads();
  return mine;
}

// The same scan for a loop of scans over ONE sumtable (k_newton_mfma) whose share per wave -- at most NB site blocks
// of RT rates -- fits the wave's registers: the first scan loads the units (and what belongs to their sites:
// weights, scaler counts, the invariant-site term), every later one finds them where they are and touches no
// memory but the model's few numbers.  Same units per wave, same order of the MFMAs and of the additions: the
// block totals are the ones deriv_block_totals produces, bit for bit.  (No per-rate scalers: those rescale the
// units per scan.)
// (NBR of the NB blocks of a wave in registers, the others in LDS behind the left operands: [block][wave][rate][k-step][lane])
template <unsigned KS, unsigned NB, unsigned NBR, unsigned RT>
struct DerivResident
{
  double2 unit[NBR ? NBR : 1][RT][KS];
  double inv_e[NB], inv_o[NB];
  SiteSide sd[NB];
};

template <unsigned KS, unsigned SREAL, unsigned NB, unsigned NBR, unsigned RT>
__device__ inline double deriv_block_totals_resident(const ModelView & mv, const ParamIdx & params, double t,
                                                     const double * sumtable,
                                                     const unsigned * ps, const unsigned * cs,
                                                     const unsigned * weights, const int * invariant,
                                                     unsigned N, unsigned nblk, double * frag,
                                                     DerivResident<KS, NB, NBR, RT> & res, bool first,
                                                     GridView gv = launch_grid())
{
  double2 * const lds_units = reinterpret_cast<double2 *>(frag + RT * KS * 64);
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const unsigned SR = SREAL ? SREAL : mv.S;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned wstride = gv.G * 4;
  const unsigned blk0 = gv.b * 4 + wave;
  if (first)
  {
#pragma unroll
    for (unsigned b = 0; b < NB; ++b)
    {
      const unsigned blk = blk0 + b * wstride;
#pragma unroll
      for (unsigned r = 0; r < RT; ++r)
#pragma unroll
        for (unsigned ks = 0; ks < KS; ++ks)
        {
          const double2 u = blk < nblk ? *reinterpret_cast<const double2 *>(sumtable + ((size_t)blk * RT + r) * UNIT + lane * 2 + ks * 128)
                                       : make_double2(0.0, 0.0);
          if (b < NBR) res.unit[b < NBR ? b : 0][r][ks] = u;
          else lds_units[((((b - NBR) * 4 + wave) * RT + r) * KS + ks) * 64 + lane] = u;
        }
    }
  }
  // left operands, as in deriv_block_totals with one trial length in all four slots
  for (unsigned x = threadIdx.x; x < RT * KS * 64; x += blockDim.x) frag[x] = 0.0;
  __syncthreads();
  for (unsigned x = threadIdx.x; x < RT * SR; x += blockDim.x)
  {
    const unsigned k = x % SR, r = x / SR;
    const unsigned pi_ = params.v[r];
    const double pinv = mv.pinv()[pi_];
    const double lam = mv.evals(pi_)[k] * mv.rates()[r] / (1.0 - pinv);
    const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
    const double ex = exp(lam * t);
    const double e0 = wr * ex, e1 = wr * ex * lam, e2 = wr * ex * lam * lam;
    double * f = frag + ((size_t)r * KS + (k >> 2)) * 64 + (k & 3) * 16;
    for (unsigned jj = 0; jj < 4; ++jj)
    {
      f[jj] = e0;
      f[jj + 4] = e1;
      f[jj + 8] = e2;
    }
  }
  __syncthreads();
  if (first)
  {
#pragma unroll
    for (unsigned b = 0; b < NB; ++b)
    {
      const unsigned blk = blk0 + b * wstride;
      const size_t site0 = (size_t)blk * S20_BS + 2 * n;
      res.sd[b] = SiteSide{0u, 0u, 0u, 0u};
      res.inv_e[b] = res.inv_o[b] = 0.0;
      if (blk >= nblk) continue;
      res.sd[b] = load_site_side(ps, cs, weights, site0, N, true);
      if (invariant)
        for (unsigned r = 0; r < RT; ++r)              // (rate by rate: the order of the additions of the streaming scan)
        {
          const unsigned pi_ = params.v[r];
          const double pinv = mv.pinv()[pi_];
          if (pinv > 0.0)
          {
            const double w = mv.weights()[r] * pinv;
            if (site0 < N && invariant[site0] >= 0) res.inv_e[b] += w * mv.freqs(pi_)[invariant[site0]];
            if (site0 + 1 < N && invariant[site0 + 1] >= 0) res.inv_o[b] += w * mv.freqs(pi_)[invariant[site0 + 1]];
          }
        }
    }
  }
  double df = 0.0, ddf = 0.0;
#pragma unroll
  for (unsigned b = 0; b < NB; ++b)
  {
    const unsigned blk = blk0 + b * wstride;
    if (blk >= nblk) break;
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    double acc_e[3] = {0.0, 0.0, 0.0}, acc_o[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (unsigned r = 0; r < RT; ++r)
    {
      const double * fr = frag + (size_t)r * KS * 64 + (lane >> 4) * 16 + (lane & 3u);     // (as in deriv_block_totals)
#pragma unroll
      for (unsigned ks = 0; ks < KS; ++ks)
      {
        const double f0 = fr[ks * 64], f1 = fr[ks * 64 + 4], f2 = fr[ks * 64 + 8];
        const double2 u = b < NBR ? res.unit[b < NBR ? b : 0][r][ks]
                                  : lds_units[((((b - NBR) * 4 + wave) * RT + r) * KS + ks) * 64 + lane];
        acc_e[0] = mfma_f64_tail(f0, u.x, acc_e[0]);
        acc_o[0] = mfma_f64_tail(f0, u.y, acc_o[0]);
        acc_e[1] = mfma_f64_tail(f1, u.x, acc_e[1]);
        acc_o[1] = mfma_f64_tail(f1, u.y, acc_o[1]);
        acc_e[2] = mfma_f64_tail(f2, u.x, acc_e[2]);
        acc_o[2] = mfma_f64_tail(f2, u.y, acc_o[2]);
      }
    }
    const SiteSide sd = res.sd[b];
    if (site0 < N)
    {
      double a = acc_e[0];
      if (res.inv_e[b] > 0.0) a += (sd.cnt_e <= 3) ? ldexp(res.inv_e[b], 256 * (int)sd.cnt_e) : INFINITY;
      const double w = (double)sd.w_e, ba = acc_e[1] / a, ca = acc_e[2] / a;
      df -= w * ba;
      ddf += w * (ba * ba - ca);
    }
    if (site0 + 1 < N)
    {
      double a = acc_o[0];
      if (res.inv_o[b] > 0.0) a += (sd.cnt_o <= 3) ? ldexp(res.inv_o[b], 256 * (int)sd.cnt_o) : INFINITY;
      const double w = (double)sd.w_o, ba = acc_o[1] / a, ca = acc_o[2] / a;
      df -= w * ba;
      ddf += w * (ba * ba - ca);
    }
    __builtin_amdgcn_sched_barrier(0);          // block by block: interleaving them costs more registers than the units take
  }
  __shared__ double part[4][8];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1)
  {
    df += __shfl_xor(df, off, 64);
    ddf += __shfl_xor(ddf, off, 64);
  }
  if (n == 0) { part[wave][2 * q] = df; part[wave][2 * q + 1] = ddf; }
  __syncthreads();
  double mine = 0.0;
  if (threadIdx.x < 8) mine = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
  __syncthreads();
  return mine;
}

template <unsigned KS, unsigned SREAL>
__global__ __launch_bounds__(256) void k_derivatives_mfma(ModelView mv, ParamIdx params, TrialLengths tl,
                                                          unsigned ntrial,
                                                          const double * sumtable,
                                                          const unsigned * ps, const unsigned * cs,
                                                          const unsigned * weights, const int * invariant,
                                                          unsigned N, unsigned nblk, unsigned R,
                                                          ReduceOut block_out, unsigned rate_scalers)
{
  extern __shared__ double frag[];        // [r][ks][lane]
  __shared__ double scratch[4];
  const double mine = deriv_block_totals<KS, SREAL>(mv, params, tl, ntrial, sumtable, ps, cs, weights, invariant,
                                                    N, nblk, R, rate_scalers, frag);
  grid_reduce_finish_lanes<8>(mine, block_out, scratch);
}

// ---------------------------------------------------------------------------
// Newton-Raphson on one branch where the data is.  The reference's minimiser evaluates {f, f'} =
// {d(-lnL)/dt, d2(-lnL)/dt2} once per iterate (src/optimize/opt_algorithms.c:133-261 calls the target function,
// src/optimize/pll_optimize.c:1223-1287, which scans the sumtable); through the C ABI every iterate is a launch,
// a completion the host waits for, and host arithmetic in between: 44 us per iterate around a 23 us scan at the
// per-GPU slice of an 8-way split.  Here ONE launch runs the whole loop: all workgroups (co-resident: the grid is
// the scan's own grid, bounded by what the chip holds at once) scan the sumtable at the current length, the block
// that draws the last ticket of the in-launch reduction adds the block totals in block order (the very sum of
// k_derivatives_mfma: same grid, same order, same bits), applies the step rule -- the expressions of the host loop
// (csrc/host/pllhip_eval.c, newton(); fp64, no contraction: -ffp-contract=off) -- and releases the next iterate
// to the others, which wait for it on a word in device memory.  The iterate trail goes to mapped host memory as it
// is made; the host waits once, for the final length.
// Every wait is bounded (NEWTON_SPIN_LIMIT polls): a workgroup that gives up marks the run as failed, so the
// grid always drains.
// ---------------------------------------------------------------------------
constexpr unsigned NEWTON_MAX_PARTS = 8;
struct NewtonControl          // device memory, one per engine
{
  double x;                   // the iterate the next scan evaluates
  double xl, xh;              // bracket
  unsigned iter;              // scans completed (what the other blocks wait for)
  unsigned status;            // NEWTON_RUNNING, or how the loop ended
  double tot[8];              // totals of the scan (grid reduction sink)
  double trail[96];           // the iterates (NEWTON_TRAIL_MAX), copied to the host when the loop ends
  // several partitions under one branch length (pllhip_newton_branch_multi): every partition runs its own launch
  // of the loop; they meet in the FIRST partition's control block
  unsigned arrived;           // partitions whose totals of the current scan are complete
  unsigned pad;
  double ptot[NEWTON_MAX_PARTS][2];   // {f, f'} of partition p at its own length s_p x
  double pscale[NEWTON_MAX_PARTS];    // s_p: the partition's branch-length scaler (1 with linked lengths)
  unsigned dbg_enter[NEWTON_MAX_PARTS], dbg_leave[NEWTON_MAX_PARTS], dbg_arrive[NEWTON_MAX_PARTS];   // PLLHIP_NEWTON_DEBUG
};

// spin_limit: polls a workgroup waits for the next iterate before it gives up (NEWTON_SPIN_LIMIT; tests lower it);
// stall_block: fault injection -- that workgroup leaves at once, as if it had never been given a CU (~0u: none)
// part / nparts / xscale: this launch is partition `part` of `nparts` that share the branch (nparts = 1: alone); it scans
// its sumtable at xscale * x and leaves its totals in ro.dst, a scratch of its own
// iter_base: the control block is not written by the host before a launch (a copy per branch saved): the loop counts its
// scans from a base no earlier launch has used (the waiters compare `iter` with iter_base + it + 1), takes the bracket
// of its first step from here (every partition leaves its scaler next to its totals), and relies on `arrived` being zero
// between loops (the last arriver of every scan resets it; a loop that was given up is followed by a memset,
// newton_finish)
struct NewtonParams
{
  double x0, bl_min, bl_max, tolerance, dxmax;
  unsigned max_newton, spin_limit, stall_block, part, nparts;
  double xscale;
  unsigned debug, iter_base;
};

// mapped host memory: [0] final length, [1] iterations, [2] status, [3] last f, [4] last df, [8 ...] the trail
constexpr unsigned NEWTON_RUNNING = 0, NEWTON_CONVERGED = 1, NEWTON_LIMIT = 2, NEWTON_NONFINITE = 3, NEWTON_STUCK = 4;
constexpr unsigned NEWTON_TRAIL_SLOT = 8, NEWTON_TRAIL_MAX = 96;
constexpr unsigned NEWTON_SPIN_LIMIT = 1u << 24;

// NB > 0: the sumtable stays in the registers of the waves between the scans (deriv_block_totals_resident; four rate
// categories, per-site scalers, at most NB blocks per wave: a 125 k-site protein slice, a 25 k-site codon slice) --
// an iterate then costs the in-launch reduction and the hand-over, not a pass over 80 MB
// what follows a scan: the block that drew the last ticket of the reduction (`last`; thread 0 finds the totals in
// ro.dst) applies the step rule and releases the next iterate; every block waits for it.  Returns the status of the
// loop after scan `it`; x = the next iterate.  s_x / s_status: two words of LDS.
__device__ inline unsigned newton_step_and_wait(unsigned it, bool last, double & x, const NewtonParams & np, const ReduceOut & ro,
                                                NewtonControl * ctl, double * host_out, unsigned long long * host_flag,
                                                unsigned long long host_seq, double * s_x, unsigned * s_status)
{
  bool apply = last && threadIdx.x == 0;
  double f = 0.0, df = 0.0;
  if (apply && np.nparts > 1)
  {
    // Several partitions share the branch: this partition's totals are complete.  They go to the shared control
    // block with the hand-over of the in-launch reductions (write-through stores, waited for, then the arrival
    // ticket; kernels_common.hpp); the partition that arrives last adds all of them in partition order with the
    // chain rule of the scalers -- f = sum s_p f_p, f' = sum s_p^2 f'_p: the host loop's derivatives()
    // (csrc/host/pllhip_eval.c; src/optimize/pll_optimize.c:1258-1267) -- and applies the step rule.
    if (np.debug) __hip_atomic_store(&ctl->dbg_arrive[np.part], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl->ptot[np.part][0], ro.dst[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl->ptot[np.part][1], ro.dst[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (... and its scaler: the host does not write the control block)
    __hip_atomic_store(&ctl->pscale[np.part], np.xscale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned k = 0; k <= REDUCE_SHARDS; ++k)                          // this partition's tickets back to zero
      __hip_atomic_store(ro.counter + k * REDUCE_SHARD_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned a = __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    apply = (a + 1 == np.nparts);
    if (apply)
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (unsigned p = 0; p < np.nparts; ++p)
      {
        const double sc = __hip_atomic_load(&ctl->pscale[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f += sc * __hip_atomic_load(&ctl->ptot[p][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        df += sc * sc * __hip_atomic_load(&ctl->ptot[p][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&ctl->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  else if (apply) { f = ro.dst[0]; df = ro.dst[1]; }
  if (apply)
  {
    // the step rule of newton() (csrc/host/pllhip_eval.c), expression by expression
    double xl = it ? ctl->xl : np.bl_min, xh = it ? ctl->xh : np.bl_max, dx;
    unsigned status = NEWTON_RUNNING;
    if (it > np.max_newton) status = NEWTON_LIMIT;                  // (the host loop counts the same way)
    else if (!isfinite(f) || !isfinite(df)) status = NEWTON_NONFINITE;
    else
    {
      if (df > 0.0)
      {
        if (fabs(f) < np.tolerance) status = NEWTON_CONVERGED;
        else
        {
          if (f < 0.0) xl = x; else xh = x;
          dx = -f / df;
        }
      }
      else
        dx = -f / fabs(df);
      if (status == NEWTON_RUNNING)
      {
        dx = fmax(fmin(dx, np.dxmax), -np.dxmax);
        if (x + dx < xl) dx = xl - x;
        if (x + dx > xh) dx = xh - x;
        if (fabs(dx) < np.tolerance) status = NEWTON_CONVERGED;
        else
        {
          x += dx;
          x = fmax(fmin(x, np.bl_max), np.bl_min);
        }
      }
    }
    // the iterate after this scan (kept on the device until the loop ends)
    if (it < NEWTON_TRAIL_MAX) ctl->trail[it] = x;
    ctl->xl = xl;
    ctl->xh = xh;
    ctl->x = x;
    ctl->status = status;
    if (np.nparts <= 1)
      for (unsigned k = 0; k <= REDUCE_SHARDS; ++k)                        // tickets back to zero for the next scan
        __hip_atomic_store(ro.counter + k * REDUCE_SHARD_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (status != NEWTON_RUNNING)
    {
      for (unsigned k = 0; k <= it && k < NEWTON_TRAIL_MAX; ++k) host_out[NEWTON_TRAIL_SLOT + k] = ctl->trail[k];
      host_out[0] = x;
      host_out[1] = (double)(it + 1);
      host_out[2] = (double)status;
      host_out[3] = f;
      host_out[4] = df;
      __threadfence_system();
      __hip_atomic_store(host_flag, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&ctl->iter, np.iter_base + it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0)
  {
    unsigned spins = 0;
    if (np.debug) __hip_atomic_fetch_add(&ctl->dbg_enter[np.part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (relaxed polls, ONE acquire afterwards: an acquire per poll invalidates the caches of the whole chip
    // several hundred times per microsecond)
    while (__hip_atomic_load(&ctl->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != np.iter_base + it + 1)
    {
      if (++spins > np.spin_limit)
      {
        // a workgroup is missing (not resident: the device is shared): the run has failed, and the host is told --
        // every block that gives up writes the same words
        __hip_atomic_store(&ctl->status, NEWTON_STUCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        host_out[0] = x;
        host_out[1] = (double)it;
        host_out[2] = (double)NEWTON_STUCK;
        __threadfence_system();
        __hip_atomic_store(host_flag, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (np.debug && spins <= np.spin_limit) __hip_atomic_fetch_add(&ctl->dbg_leave[np.part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_status = (spins > np.spin_limit) ? NEWTON_STUCK
                                        : __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_x = __hip_atomic_load(&ctl->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned status = *s_status;
  x = *s_x;
  __syncthreads();
  return status;
}

template <unsigned KS, unsigned SREAL, unsigned NB, unsigned NBR = NB>
__device__ inline void newton_loop(const ModelView & mv, const ParamIdx & params, const NewtonParams & np,
                                   const double * sumtable,
                                   const unsigned * ps, const unsigned * cs,
                                   const unsigned * weights, const int * invariant,
                                   unsigned N, unsigned nblk, unsigned R,
                                   const ReduceOut & ro, unsigned rate_scalers,
                                   NewtonControl * ctl, double * host_out,
                                   unsigned long long * host_flag, unsigned long long host_seq,
                                   GridView gv = launch_grid())
{
  extern __shared__ double frag[];
  __shared__ double scratch[4];
  __shared__ double s_x;
  __shared__ unsigned s_status;
  double x = np.x0;                                     // (clamped by the host, as newton() does first)
  if (gv.b == np.stall_block) return;                   // (fault injection: a workgroup that never arrives)
  DerivResident<KS, NB ? NB : 1, NB ? NBR : 1, NB ? 4 : 1> res;
  for (unsigned it = 0; ; ++it)
  {
    double mine;
    if constexpr (NB > 0)
      mine = deriv_block_totals_resident<KS, SREAL, NB, NBR, 4>(mv, params, np.xscale * x, sumtable, ps, cs, weights, invariant, N, nblk, frag, res, it == 0, gv);
    else
    {
      TrialLengths tl;
#pragma unroll
      for (unsigned i = 0; i < MAX_TRIAL_LENGTHS; ++i) tl.t[i] = np.xscale * x;
      mine = deriv_block_totals<KS, SREAL>(mv, params, tl, 1u, sumtable, ps, cs, weights, invariant,
                                           N, nblk, R, rate_scalers, frag, gv);
    }
    const bool last = grid_reduce_finish_lanes<8, true>(mine, ro, scratch, gv);
    if (newton_step_and_wait(it, last, x, np, ro, ctl, host_out, host_flag, host_seq, &s_x, &s_status) != NEWTON_RUNNING) return;
  }
}

#define PLLHIP_NEWTON_ARGS ModelView mv, ParamIdx params, NewtonParams np, const double * sumtable,                  \
                           const unsigned * ps, const unsigned * cs, const unsigned * weights, const int * invariant, \
                           unsigned N, unsigned nblk, unsigned R, ReduceOut ro, unsigned rate_scalers,               \
                           NewtonControl * ctl, double * host_out, unsigned long long * host_flag, unsigned long long host_seq
template <unsigned KS, unsigned SREAL>
__global__ __launch_bounds__(256) void k_newton_mfma(PLLHIP_NEWTON_ARGS)
{
  newton_loop<KS, SREAL, 0>(mv, params, np, sumtable, ps, cs, weights, invariant, N, nblk, R, ro, rate_scalers, ctl, host_out, host_flag, host_seq);
}

// one wave per SIMD with the whole register file: four blocks of five k-steps (20 states: 131 k sites on 256 CUs;
// NBR = 3 of them in registers, one in LDS: four in registers spill 104) or one block of sixteen (33 .. 64 states:
// 32 k sites).  dynamic LDS = the left operands + (NB - NBR) * 4 waves * 4 rates * KS * 64 double2
template <unsigned KS, unsigned SREAL, unsigned NB, unsigned NBR>
__global__ __launch_bounds__(256, 1) void k_newton_mfma_resident(PLLHIP_NEWTON_ARGS)
{
  newton_loop<KS, SREAL, NB, NBR>(mv, params, np, sumtable, ps, cs, weights, invariant, N, nblk, R, ro, rate_scalers, ctl, host_out, host_flag, host_seq);
}
#undef PLLHIP_NEWTON_ARGS

// ---------------------------------------------------------------------------
// layout converters between the API layout [site][rate][20] and the blocked
// device layout (host materialisation, tip CLV upload, checkpoint restore)
// ---------------------------------------------------------------------------
// (shared by the 61-state family: `rows` = states_padded = rows per unit)
__global__ __launch_bounds__(256) void k_s20_to_blocked(const double * api, double * blocked,
                                                        unsigned N, unsigned nblk, unsigned R,
                                                        unsigned Sp, unsigned rows)
{
  // API rows have stride Sp (states_padded); a blocked unit has `rows` >= S state rows
  const unsigned unit = rows * S20_BS;
  const unsigned long long total = (unsigned long long)nblk * R * unit;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * blockDim.x)
  {
    const unsigned s = e % S20_BS, j = (e / S20_BS) % rows;
    const unsigned long long br = e / unit;
    const unsigned r = br % R;
    const unsigned long long n = (br / R) * S20_BS + s;
    blocked[e] = (n < N && j < Sp) ? api[(n * R + r) * Sp + j] : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_s20_from_blocked(const double * blocked, double * api,
                                                          unsigned N, unsigned R, unsigned Sp, unsigned rows)
{
  const unsigned long long total = (unsigned long long)N * R * Sp;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * blockDim.x)
  {
    const unsigned j = e % Sp, r = (e / Sp) % R;
    const unsigned long long n = e / Sp / R;
    api[e] = blocked[(((n / S20_BS) * R + r) * rows + j) * S20_BS + (n % S20_BS)];
  }
}

// --- launchers -------------------------------------------------------------

static unsigned s20_grid(const Engine * e, unsigned blocks_per_cu)
{
  static const int env_bpc = getenv("PLLHIP_S20_BPC") ? atoi(getenv("PLLHIP_S20_BPC")) : 0;
  if (env_bpc > 0) blocks_per_cu = (unsigned)env_bpc;
  const unsigned need = (e->nblk + 3) / 4;
  return std::max(1u, std::min(need, e->cu_count * blocks_per_cu));
}

static int launch_partials_s20(Engine * e, const OpBatch & batch, unsigned nops)
{
  const size_t lds = sizeof(double) * 2 * e->R * S20_FRAGS;
  static const int unroll = getenv("PLLHIP_S20_UNROLL") ? atoi(getenv("PLLHIP_S20_UNROLL")) : 0;
  static const unsigned env_flags = getenv("PLLHIP_S20_NT") ? (unsigned)atoi(getenv("PLLHIP_S20_NT")) & 3u : 0u;
  const unsigned flags = env_flags | (e->rate_scalers ? 4u : 0u);
  if (e->R == 4 && !unroll)    // PLLHIP_S20_UNROLL=1 selects the runtime-R variant for A/B runs
    hipLaunchKernelGGL(k_partials_s20<4>, dim3(s20_grid(e, 4), nops), dim3(256), lds, e->stream,
                       batch, e->nblk, e->R, e->lut_codes, flags);
  else
    hipLaunchKernelGGL(k_partials_s20<0>, dim3(s20_grid(e, 4), nops), dim3(256), lds, e->stream,
                       batch, e->nblk, e->R, e->lut_codes, flags);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static bool chains_supported_s20(const Engine * e) { return e->R == 4 || e->R == 2 || e->R == 1; }

// LDS doubles of the two children's tables of an operation in a chain (0: read from global)
static bool s20_chain_lut_lds(const Engine * e, unsigned lut_used)
{
  return e->R * lut_used * S20_LUT_RS <= 2560u;       // keeps a tip table at most as large as before
}

static unsigned s20_chain_slot(const Engine * e, bool tip, unsigned lut_used)
{
  if (!tip) return e->R * S20_CFRAGS;
  return s20_chain_lut_lds(e, lut_used) ? ((e->R * lut_used * S20_LUT_RS + 7u) & ~7u) : 0u;
}

// (A second geometry for small slices -- chains of two, two 256-thread workgroups per CU, so
// that one workgroup's fragment fill overlaps the other's streaming -- was measured at
// 125 k / 250 k / 500 k sites and is 13 % / 15 % / 0 % slower than this one: the extra
// re-reads cost more than the overlap gains.)
// every instantiation of a chain kernel may use the whole LDS
template <class K>
static int s20_allow_full_lds(K kernel)
{
  PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(double) * S20_CHAIN_LDS)));
  return PLL_SUCCESS;
}

// R in {1, 2, 4} x per-site / per-rate scalers
#define PLLHIP_S20_CHAIN_DISPATCH(KERNEL, CALL)                        \
  do {                                                                 \
    if (e->rate_scalers)                                               \
    {                                                                  \
      if (e->R == 4) { CALL((KERNEL<4, true>)); }                      \
      else if (e->R == 2) { CALL((KERNEL<2, true>)); }                 \
      else { CALL((KERNEL<1, true>)); }                                \
    }                                                                  \
    else                                                               \
    {                                                                  \
      if (e->R == 4) { CALL((KERNEL<4, false>)); }                     \
      else if (e->R == 2) { CALL((KERNEL<2, false>)); }                \
      else { CALL((KERNEL<1, false>)); }                               \
    }                                                                  \
  } while (0)

static int launch_chains_s20(Engine * e, const ChainBatch & batch, unsigned nchains, unsigned lds_doubles,
                             unsigned lut_used)
{
  const size_t lds = sizeof(double) * lds_doubles;
  const unsigned env_flags = []() { const char * v = getenv("PLLHIP_S20_NT"); return v ? (unsigned)atoi(v) & 3u : 0u; }();
  const unsigned flags = env_flags | (s20_chain_lut_lds(e, lut_used) ? 8u : 0u);
  static bool attr_set_dev[64] = {false};          // per device: one process may drive several GPUs
  bool & attr_set = attr_set_dev[e->device & 63];
  if (!attr_set)
  {
    if (!s20_allow_full_lds(k_chain_s20<4, false>) || !s20_allow_full_lds(k_chain_s20<2, false>) ||
        !s20_allow_full_lds(k_chain_s20<1, false>) || !s20_allow_full_lds(k_chain_s20<4, true>) ||
        !s20_allow_full_lds(k_chain_s20<2, true>) || !s20_allow_full_lds(k_chain_s20<1, true>))
      return PLL_FAILURE;
    attr_set = true;
  }
  // workgroups per CU and chain (measured at 1 M sites: 1 beats 2 and 4)
  static const int env_bpc = getenv("PLLHIP_S20_CHAIN_BPC") ? atoi(getenv("PLLHIP_S20_CHAIN_BPC")) : 1;
  const unsigned need = (e->nblk + S20_CHAIN_WAVES - 1) / S20_CHAIN_WAVES;
  const unsigned gx = std::max(1u, std::min(need, e->cu_count * (unsigned)std::max(1, env_bpc)));
  const dim3 grid(gx, nchains), block(64 * S20_CHAIN_WAVES);
#define PLLHIP_CALL(K) hipLaunchKernelGGL(K, grid, block, lds, e->stream, batch, e->nblk, e->lut_codes, lut_used, flags)
  PLLHIP_S20_CHAIN_DISPATCH(k_chain_s20, PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// `extent`: site blocks of the largest partition the chains [chain_begin, chain_end) belong to
static int launch_traverse_s20(Engine * e, const PlanView & plan, unsigned lds_doubles, unsigned extent,
                               unsigned chain_begin, unsigned chain_end, unsigned rows, unsigned row_wgs_per_cu = 0,
                               bool wide = false, bool transient = false)
{
  const size_t lds = sizeof(double) * lds_doubles;
  const unsigned flags = []() { const char * v = getenv("PLLHIP_S20_NT"); return v ? (unsigned)atoi(v) & 3u : 0u; }();
  static bool attr_set_dev[64] = {false};
  bool & attr_set = attr_set_dev[e->device & 63];
  if (!attr_set)
  {
#define PLLHIP_ALLOW(RS_, W_, T_) (s20_allow_full_lds(k_traverse_s20<4, RS_, W_, T_>) && s20_allow_full_lds(k_traverse_s20<2, RS_, W_, T_>) && \
                                   s20_allow_full_lds(k_traverse_s20<1, RS_, W_, T_>))
    if (!PLLHIP_ALLOW(false, false, false) || !PLLHIP_ALLOW(true, false, false) || !PLLHIP_ALLOW(false, true, false) ||
        !PLLHIP_ALLOW(false, false, true) || !PLLHIP_ALLOW(true, false, true) || !PLLHIP_ALLOW(false, true, true))
      return PLL_FAILURE;
#undef PLLHIP_ALLOW
    attr_set = true;
  }
  const unsigned need = (extent + S20_CHAIN_WAVES - 1) / S20_CHAIN_WAVES;
  // (a row keeps at least the workgroups that leave a wave 16 site blocks: large partitions are shared out finely --
  // 256 instead of 26 workgroups per row in the first round of C3, 32.9 against 33.7 ms on one box)
  const unsigned gx = round_grid(e, std::max(1u, std::min(need, e->cu_count)), rows, row_wgs_per_cu ? row_wgs_per_cu : 4u,
                                 (extent + 16u * S20_CHAIN_WAVES - 1) / (16u * S20_CHAIN_WAVES));
  // slab: site blocks that go through the whole schedule together (default: all of them; smaller
  // slabs were measured and lose to the per-chain re-staging)
  static const int env_slab = getenv("PLLHIP_S20_SLAB") ? atoi(getenv("PLLHIP_S20_SLAB")) : 0;
  const unsigned per_pass = gx * S20_CHAIN_WAVES;
  unsigned slab = env_slab > 0 ? (unsigned)env_slab : extent;
  slab = std::max(per_pass, (slab + per_pass - 1) / per_pass * per_pass);   // whole passes of the grid
  const dim3 grid(gx, std::max(1u, rows)), block(64 * S20_CHAIN_WAVES);
#define PLLHIP_CALL(K) hipLaunchKernelGGL(K, grid, block, lds, e->stream, plan, chain_begin, chain_end, extent, slab, flags)
#define PLLHIP_BY_RATES(RS_, W_, T_)                                              \
  do {                                                                           \
    if (e->R == 4) { PLLHIP_CALL((k_traverse_s20<4, RS_, W_, T_>)); }             \
    else if (e->R == 2) { PLLHIP_CALL((k_traverse_s20<2, RS_, W_, T_>)); }        \
    else { PLLHIP_CALL((k_traverse_s20<1, RS_, W_, T_>)); }                       \
  } while (0)
  // (wide tips -- site repeats, tips kept per class -- exist with per-site scaling only)
  if (wide && !e->rate_scalers) { if (transient) PLLHIP_BY_RATES(false, true, true); else PLLHIP_BY_RATES(false, true, false); }
  else if (e->rate_scalers) { if (transient) PLLHIP_BY_RATES(true, false, true); else PLLHIP_BY_RATES(true, false, false); }
  else { if (transient) PLLHIP_BY_RATES(false, false, true); else PLLHIP_BY_RATES(false, false, false); }
#undef PLLHIP_BY_RATES
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_edge_lnl_s20(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                               const NodeRef & parent, const NodeRef & child,
                               const double * pm, const double * lut,
                               const unsigned * ps, const unsigned * cs,
                               double * persite, unsigned nblocks)
{
  const size_t lds = sizeof(double) * e->R * S20_FRAGS;
  hipLaunchKernelGGL(k_edge_lnl_s20, dim3(nblocks), dim3(256), lds, e->stream,
                     mv, fidx, parent, child, pm, lut, e->lut_codes, ps, cs,
                     e->d_weights, e->d_invariant, e->d_tipmap, e->N, e->nblk, e->R,
                     persite, reduce_out(e), e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_sumtable_s20(Engine * e, const ModelView & mv, const ParamIdx & params,
                               const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  // scratch: Lm | Rm | lutL | lutR
  const size_t mats = (size_t)e->R * 400, luts = (size_t)e->R * std::max(1u, e->lut_codes) * 20;
  if (!e->d_sum_scratch)
  {
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&e->d_sum_scratch),
                               sizeof(double) * 2 * (mats + (size_t)e->R * PLL_ASCII_SIZE * 20));
    if (err != hipSuccess)
    {
      set_error(PLL_ERROR_MEM_ALLOC, "hipMalloc for sumtable scratch failed");
      return PLL_FAILURE;
    }
  }
  double * Lm = e->d_sum_scratch, * Rm = Lm + mats, * lutL = Rm + mats, * lutR = lutL + luts;
  const bool want_lut = parent.codes || child.codes;
  if (sum_prep_needed(e, params, want_lut))
  {
    hipLaunchKernelGGL(k_sumtable_prep_s20, dim3(e->R), dim3(256), 0, e->stream,
                       mv, params, e->d_tipmap, e->lut_codes, want_lut, Lm, Rm, lutL, lutR);
    PLLHIP_TRY(hipGetLastError());
  }
  OpBatch batch;
  OpDesc & d = batch.op[0];
  d.clv1 = parent.clv; d.codes1 = parent.codes; d.pmat1 = Lm; d.lut1 = lutL;
  d.clv2 = child.clv;  d.codes2 = child.codes;  d.pmat2 = Rm; d.lut2 = lutR;
  d.scaler1 = d.scaler2 = nullptr;
  d.parent = d_sum;
  d.parent_scaler = nullptr;
  return launch_partials_s20(e, batch, 1);
}

// up to 4 trial lengths per launch (spare rows repeat the last length)
static int launch_derivatives_s20(Engine * e, const ModelView & mv, const ParamIdx & params,
                                  const TrialLengths & tl, unsigned count,
                                  const double * d_sum, const unsigned * ps, const unsigned * cs,
                                  unsigned nblocks)
{
  hipLaunchKernelGGL((k_derivatives_mfma<5, 20>), dim3(nblocks), dim3(256), sizeof(double) * e->R * 5 * 64, e->stream,
                     mv, params, tl, count, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->nblk, e->R, reduce_out(e),
                     e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

} // namespace pllhip
