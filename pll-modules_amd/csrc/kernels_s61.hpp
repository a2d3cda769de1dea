// kernels_s61.hpp -- 33 .. 64 states on the fp64 matrix cores: the 61-state codon models it was written for
// (state count a compile-time constant there), the 60 / 62 / 63-codon genetic codes and multistate alphabets up
// to 64 states (src/util/models_mult.c:92-97) with the state count as a run-time value S (P-matrix rows Sp
// apart); units always have 64 rows.
//
// Same blocked device layout and lane mapping as the 20-state family
// (kernels_s20.hpp): clv[site_block][rate][state row][32 sites], a (block, rate)
// unit is a 64 x 32 fp64 matrix (16 KiB; rows 61..63 are zero padding), lane
// l = 16q + n holds sites 2n, 2n+1, and slot k of a lane is row 4k + q -- for
// the MFMA B operand (k-step k) and for the MFMA D result (register k%4 of
// M-tile k/4) alike.  Every CLV load/store is a fully coalesced 1 KiB wave
// instruction.
//
// What differs from 20 states is the balance: a site-update moves 1 467 B for
// 14 945 flops (AI 10.2 flop/B), i.e. the kernel sits on the ridge
// between HBM and the FP64 matrix pipe (256 MFMA = 16 384 SIMD cycles per unit
// against ~4 700 CU cycles of HBM time per unit), and has to keep both busy at
// once.  The A fragments of ONE rate (both children: 2 x 4 M-tiles x 16 k-steps x
// 512 B = 64 KiB) fill the LDS, so a workgroup walks the rates one after the other
// over its range of site blocks.  Results are stored unscaled; the per-site scaling
// vote (all R*61 entries) is known after the last rate and the rare rescale is a
// fix-up pass over the units just written.
#pragma once

#include "kernels_common.hpp"
#include "kernels_s20.hpp"
#include "engine.h"

namespace pllhip {

constexpr unsigned S61_S = 61;
constexpr unsigned S61_SP = 64;
constexpr unsigned S61_KS = 16;                      // k-steps = slots per lane
constexpr unsigned S61_MT = 4;                       // 16-row M tiles
constexpr unsigned S61_UNIT = S61_SP * S20_BS;       // doubles per (block, rate) unit
constexpr unsigned S61_FRAGS = S61_MT * S61_KS * 64; // A-fragment doubles per (child, rate)

// A fragments of rate r of a [R][61][64] row-major matrix set:
//   frag[(mt*16 + ks)*64 + lane] = M[r][(lane&15) + 16*mt][4*ks + (lane>>4)]   (0 beyond row/col 60)
__device__ inline void s61_fill_frags(double * frag, const double * mats, unsigned r, unsigned S, unsigned Sp)
{
  const double * M = mats + (size_t)r * S * Sp;
  staged_loop<16>(S61_FRAGS, [=](unsigned e)
  {
    const unsigned lane = e & 63, f = e >> 6, ks = f & 15, mt = f >> 4;
    const unsigned i = (lane & 15) + 16 * mt, j = 4 * ks + (lane >> 4);
    const bool in = i < S && j < S;
    const double x = M[in ? (size_t)i * Sp + j : 0];
    return in ? x : 0.0;
  }, [=](unsigned e, double x) { frag[e] = x; });
}

// child term in D layout: t[k] = {even site, odd site} for row 4k + q
__device__ inline void s61_child_inner(const double * unit, const double * frag, unsigned lane,
                                       double2 t[S61_KS])
{
  const unsigned off = lane * 2;
  v4d acc[S61_MT][2];
#pragma unroll
  for (unsigned mt = 0; mt < S61_MT; ++mt)
  {
    acc[mt][0] = v4d{0, 0, 0, 0};
    acc[mt][1] = v4d{0, 0, 0, 0};
  }
#pragma unroll
  for (unsigned half = 0; half < 2; ++half)
  {
    double2 b[8];
#pragma unroll
    for (unsigned k = 0; k < 8; ++k)
      b[k] = *reinterpret_cast<const double2 *>(unit + (half * 8 + k) * 128 + off);
#pragma unroll
    for (unsigned k = 0; k < 8; ++k)
    {
      const unsigned ks = half * 8 + k;
#pragma unroll
      for (unsigned mt = 0; mt < S61_MT; ++mt)
      {
        const double a = frag[(mt * S61_KS + ks) * 64 + lane];
        acc[mt][0] = mfma_f64(a, b[k].x, acc[mt][0]);
        acc[mt][1] = mfma_f64(a, b[k].y, acc[mt][1]);
      }
    }
  }
#pragma unroll
  for (unsigned mt = 0; mt < S61_MT; ++mt)
#pragma unroll
    for (unsigned v = 0; v < 4; ++v)
      t[mt * 4 + v] = make_double2(acc[mt][0][v], acc[mt][1][v]);
}

// LUT row layout [code][61]
__device__ inline void s61_child_tip(const double * lut_r, unsigned code_e, unsigned code_o,
                                     unsigned q, double2 t[S61_KS], unsigned S)
{
  const double * le = lut_r + code_e * S, * lo = lut_r + code_o * S;
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
  {
    const unsigned i = 4 * k + q;
    t[k] = (i < S) ? make_double2(le[i], lo[i]) : make_double2(0.0, 0.0);
  }
}

__device__ inline void s61_load_d(const double * unit, unsigned lane, double2 t[S61_KS])
{
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
    t[k] = *reinterpret_cast<const double2 *>(unit + k * 128 + lane * 2);
}

__device__ inline void s61_store_d(double * unit, unsigned lane, const double2 t[S61_KS])
{
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
    *reinterpret_cast<double2 *>(unit + k * 128 + lane * 2) = t[k];
}

__device__ inline void s61_tip_d(unsigned long long mask_e, unsigned long long mask_o, unsigned q,
                                 double2 t[S61_KS], unsigned S)
{
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
  {
    const unsigned i = 4 * k + q;
    t[k] = (i < S) ? make_double2((double)((mask_e >> i) & 1ULL), (double)((mask_o >> i) & 1ULL))
                       : make_double2(0.0, 0.0);
  }
}

// ---------------------------------------------------------------------------
// partials (also the sumtable, with eigen-basis matrices).  Every workgroup owns ONE
// contiguous range of site blocks and walks it once per rate, so the A fragments of a
// rate (both children: 64 KiB of LDS, or the tip lookup table of the rate when it fits:
// codes * 61 <= 4096) are filled once per workgroup and rate.  The B operands are
// fetched half a child (8 KiB per wave) ahead of the MFMAs that consume them, into two
// alternating register sets, so a wave always has loads in flight while it (or the
// other wave of its SIMD) feeds the matrix pipe.  (Measured against the first version
// of this kernel -- per-tile fragment fill, loads issued right before their MFMAs --
// this is 1.2x faster at 200k sites and 2.7x at 25k; one wave per SIMD with whole-unit
// prefetch was tried and is no faster.)
// Fragment layout (pairs of M tiles, one ds_read_b128 feeds four MFMAs):
//   frag[((ks*2 + mp)*64 + lane)*2 + h] = M[r][(lane&15) + 16*(2*mp + h)][4*ks + (lane>>4)]
// grid = (<= 2 x CUs, ops), block = 256, dynamic LDS = 64 KiB.
// ---------------------------------------------------------------------------
constexpr unsigned S61_CHUNK = 128;      // blocks per workgroup pass: 4 waves x 32 flag bits

__device__ inline void s61_fill_frags_v3(double * frag, const double * mats, unsigned r, unsigned S, unsigned Sp)
{
  const double * M = mats + (size_t)r * S * Sp;
  staged_loop<16>(S61_FRAGS, [=](unsigned e)
  {
    const unsigned h = e & 1, lane = (e >> 1) & 63, f = e >> 7, mp = f & 1, ks = f >> 1;
    const unsigned i = (lane & 15) + 16 * (2 * mp + h), j = 4 * ks + (lane >> 4);
    const bool in = i < S && j < S;
    const double x = M[in ? (size_t)i * Sp + j : 0];
    return in ? x : 0.0;
  }, [=](unsigned e, double x) { frag[e] = x; });
}

#define S61_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <unsigned HALF>
__device__ inline void s61_issue_half(const double * unit, unsigned lane, double2 b[8])
{
#pragma unroll
  for (unsigned k = 0; k < 8; ++k)
    b[k] = *reinterpret_cast<const double2 *>(unit + (HALF * 8 + k) * 128 + lane * 2);
}

template <unsigned HALF>
__device__ inline void s61_mfma_half(const double2 b[8], const double2 * fragv, unsigned lane,
                                     v4d acc[S61_MT][2])
{
#pragma unroll
  for (unsigned k = 0; k < 8; ++k)
  {
    const unsigned ks = HALF * 8 + k;
#pragma unroll
    for (unsigned mp = 0; mp < 2; ++mp)
    {
      const double2 a = fragv[(ks * 2 + mp) * 64 + lane];
      acc[2 * mp][0] = mfma_f64(a.x, b[k].x, acc[2 * mp][0]);
      acc[2 * mp][1] = mfma_f64(a.x, b[k].y, acc[2 * mp][1]);
      acc[2 * mp + 1][0] = mfma_f64(a.y, b[k].x, acc[2 * mp + 1][0]);
      acc[2 * mp + 1][1] = mfma_f64(a.y, b[k].y, acc[2 * mp + 1][1]);
    }
  }
}

__device__ inline void s61_acc_zero(v4d acc[S61_MT][2])
{
#pragma unroll
  for (unsigned mt = 0; mt < S61_MT; ++mt)
  {
    acc[mt][0] = v4d{0, 0, 0, 0};
    acc[mt][1] = v4d{0, 0, 0, 0};
  }
}

// product of the two child terms, scaling vote, store: t1 * acc -> parent unit
// fe / fo: power-of-two factors the unit is stored with (the PREDICTED scaling of its two
// sites in rate-parallel launches, 1 otherwise); the vote is taken on the unscaled values
__device__ inline void s61_finish_unit(double * dst, unsigned lane, unsigned q, double2 t1[S61_KS],
                                       const double2 t2[S61_KS], unsigned bit, unsigned & small_e,
                                       unsigned & small_o, double fe = 1.0, double fo = 1.0)
{
  int se = 1, so = 1;
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
  {
    t1[k].x *= t2[k].x;
    t1[k].y *= t2[k].y;
    se &= (t1[k].x < SCALE_THRESHOLD);        // rows >= S are zero (padded fragments and table rows): they never veto
    so &= (t1[k].y < SCALE_THRESHOLD);
    t1[k].x *= fe;
    t1[k].y *= fo;
  }
  if (!se) small_e &= ~bit;
  if (!so) small_o &= ~bit;
  s61_store_d(dst, lane, t1);
}

// PLL_ATTRIB_RATE_SCALERS: the vote covers this unit alone -- decide (all four lane groups),
// scale, store, and write the count of (site, rate): no votes, no prediction, no fix-up kernel
__device__ inline void s61_finish_unit_rs(const OpDesc & op, unsigned blk, unsigned r, unsigned R,
                                          double * dst, unsigned lane, double2 t1[S61_KS], const double2 t2[S61_KS])
{
  const unsigned q = lane >> 4, n = lane & 15;
  int se = 1, so = 1;
#pragma unroll
  for (unsigned k = 0; k < S61_KS; ++k)
  {
    t1[k].x *= t2[k].x;
    t1[k].y *= t2[k].y;
    se &= (t1[k].x < SCALE_THRESHOLD);        // rows >= S are zero: they never veto
    so &= (t1[k].y < SCALE_THRESHOLD);
  }
  if (op.parent_scaler)
  {
    se = s20_and_q(se);
    so = s20_and_q(so);
    const double fe = se ? SCALE_FACTOR : 1.0, fo = so ? SCALE_FACTOR : 1.0;
#pragma unroll
    for (unsigned k = 0; k < S61_KS; ++k) { t1[k].x *= fe; t1[k].y *= fo; }
    if (q == 0)
    {
      const size_t site0 = (size_t)blk * S20_BS + 2 * n;
      const size_t xe = site0 * R + r, xo = (site0 + 1) * R + r;
      unsigned ce = se ? 1u : 0u, co = so ? 1u : 0u;
      if (op.scaler1) { ce += op.scaler1[xe]; co += op.scaler1[xo]; }
      if (op.scaler2) { ce += op.scaler2[xe]; co += op.scaler2[xo]; }
      op.parent_scaler[xe] = ce;
      op.parent_scaler[xo] = co;
    }
  }
  s61_store_d(dst, lane, t1);
}

__device__ inline void s61_pred_factors(const uint8_t * pred, unsigned blk, unsigned lane, double & fe, double & fo)
{
  fe = fo = 1.0;
  if (pred)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * (lane & 15);
    fe = pred[site0] ? SCALE_FACTOR : 1.0;
    fo = pred[site0 + 1] ? SCALE_FACTOR : 1.0;
  }
}

__device__ inline void s61_acc_to_t(const v4d acc[S61_MT][2], double2 t[S61_KS])
{
#pragma unroll
  for (unsigned mt = 0; mt < S61_MT; ++mt)
#pragma unroll
    for (unsigned v = 0; v < 4; ++v)
      t[mt * 4 + v] = make_double2(acc[mt][0][v], acc[mt][1][v]);
}

// one rate of a wave's blocks (first, first + W, ...: nb of them; W = waves of the workgroup), both children inner
__device__ inline void s61_rate_ii(const OpDesc & op, const double * frag, const double * frag2,
                                   unsigned r, unsigned R, unsigned first, unsigned nb, unsigned lane,
                                   unsigned & small_e, unsigned & small_o, const uint8_t * pred, bool rs, unsigned W = 4)
{
  const double2 * f1 = reinterpret_cast<const double2 *>(frag);
  const double2 * f2 = reinterpret_cast<const double2 *>(frag2);
  const unsigned q = lane >> 4;
  double2 bA[8], bB[8];
  s61_issue_half<0>(op.clv1 + ((size_t)first * R + r) * S61_UNIT, lane, bA);
  for (unsigned i = 0; i < nb; ++i)
  {
    const size_t ub = ((size_t)(first + W * i) * R + r) * S61_UNIT;
    const size_t ubn = ((size_t)(first + W * (i + 1 < nb ? i + 1 : i)) * R + r) * S61_UNIT;
    v4d acc[S61_MT][2];
    double2 t1[S61_KS], t2[S61_KS];
    s61_acc_zero(acc);
    s61_issue_half<1>(op.clv1 + ub, lane, bB);
    S61_SCHED_FENCE();
    s61_mfma_half<0>(bA, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_issue_half<0>(op.clv2 + ub, lane, bA);
    S61_SCHED_FENCE();
    s61_mfma_half<1>(bB, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_acc_to_t(acc, t1);
    s61_acc_zero(acc);
    s61_issue_half<1>(op.clv2 + ub, lane, bB);
    S61_SCHED_FENCE();
    s61_mfma_half<0>(bA, f2, lane, acc);
    S61_SCHED_FENCE();
    s61_issue_half<0>(op.clv1 + ubn, lane, bA);      // next block (the last one re-reads itself)
    S61_SCHED_FENCE();
    s61_mfma_half<1>(bB, f2, lane, acc);
    S61_SCHED_FENCE();
    s61_acc_to_t(acc, t2);
    if (rs) { s61_finish_unit_rs(op, first + W * i, r, R, op.parent + ub, lane, t1, t2); continue; }
    double fe, fo;
    s61_pred_factors(pred, first + W * i, lane, fe, fo);
    s61_finish_unit(op.parent + ub, lane, q, t1, t2, 1u << i, small_e, small_o, fe, fo);
  }
}

// one inner child (clv, fragments) and one tip child (codes, per-rate table in LDS or global)
__device__ inline void s61_rate_ti(const OpDesc & op, const double * clv, const double * fragi,
                                   const unsigned char * codes, const double * lut_r,
                                   unsigned r, unsigned R, unsigned first, unsigned nb, unsigned lane,
                                   unsigned & small_e, unsigned & small_o, const uint8_t * pred, bool rs, unsigned S, unsigned W = 4)
{
  const double2 * f1 = reinterpret_cast<const double2 *>(fragi);
  const unsigned q = lane >> 4, n = lane & 15;
  double2 bA[8], bB[8];
  s61_issue_half<0>(clv + ((size_t)first * R + r) * S61_UNIT, lane, bA);
  for (unsigned i = 0; i < nb; ++i)
  {
    const unsigned blk = first + W * i;
    const size_t ub = ((size_t)blk * R + r) * S61_UNIT;
    const size_t ubn = ((size_t)(first + W * (i + 1 < nb ? i + 1 : i)) * R + r) * S61_UNIT;
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    const unsigned ce = codes[site0], co = codes[site0 + 1];
    v4d acc[S61_MT][2];
    double2 t1[S61_KS], t2[S61_KS];
    s61_acc_zero(acc);
    s61_issue_half<1>(clv + ub, lane, bB);
    S61_SCHED_FENCE();
    s61_mfma_half<0>(bA, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_issue_half<0>(clv + ubn, lane, bA);
    S61_SCHED_FENCE();
    s61_mfma_half<1>(bB, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_acc_to_t(acc, t1);
    s61_child_tip(lut_r, ce, co, q, t2, S);
    if (rs) { s61_finish_unit_rs(op, blk, r, R, op.parent + ub, lane, t1, t2); continue; }
    double fe, fo;
    s61_pred_factors(pred, blk, lane, fe, fo);
    s61_finish_unit(op.parent + ub, lane, q, t1, t2, 1u << i, small_e, small_o, fe, fo);
  }
}

__device__ inline void s61_rate_tt(const OpDesc & op, const double * lut1_r, const double * lut2_r,
                                   unsigned r, unsigned R, unsigned first, unsigned nb, unsigned lane,
                                   unsigned & small_e, unsigned & small_o, const uint8_t * pred, bool rs, unsigned S, unsigned W = 4)
{
  const unsigned q = lane >> 4, n = lane & 15;
  // the tip codes of block i+1 are fetched while block i is looked up and stored
  size_t site0 = (size_t)first * S20_BS + 2 * n;
  unsigned c1e = op.codes1[site0], c1o = op.codes1[site0 + 1];
  unsigned c2e = op.codes2[site0], c2o = op.codes2[site0 + 1];
  for (unsigned i = 0; i < nb; ++i)
  {
    const unsigned blk = first + W * i;
    const size_t siten = (size_t)(first + W * (i + 1 < nb ? i + 1 : i)) * S20_BS + 2 * n;
    const unsigned n1e = op.codes1[siten], n1o = op.codes1[siten + 1];
    const unsigned n2e = op.codes2[siten], n2o = op.codes2[siten + 1];
    double2 t1[S61_KS], t2[S61_KS];
    s61_child_tip(lut1_r, c1e, c1o, q, t1, S);
    s61_child_tip(lut2_r, c2e, c2o, q, t2, S);
    if (rs) s61_finish_unit_rs(op, blk, r, R, op.parent + ((size_t)blk * R + r) * S61_UNIT, lane, t1, t2);
    else
    {
      double fe, fo;
      s61_pred_factors(pred, blk, lane, fe, fo);
      s61_finish_unit(op.parent + ((size_t)blk * R + r) * S61_UNIT, lane, q, t1, t2, 1u << i,
                      small_e, small_o, fe, fo);
    }
    c1e = n1e; c1o = n1o; c2e = n2e; c2o = n2o;
  }
}

// Second half of a rate-parallel launch: combine the R votes of every site of blocks beg, beg + nwaves, ... < end,
// correct the units of the sites whose vote differs from the prediction they were stored with (a vector is usually
// re-evaluated many times -- branch-length optimisation, SPR scoring -- and its scaling pattern rarely changes),
// write the parent scalers and the new prediction.  One wave per block.
__device__ inline void s61_fixup_range(const OpDesc & op, unsigned opi, unsigned beg, unsigned end, unsigned nwaves,
                                       unsigned nblk, unsigned R, const uint8_t * votes, const uint8_t * pred_in,
                                       uint8_t * pred_out, unsigned lane)
{
  const unsigned q = lane >> 4, n = lane & 15;
  const size_t nsite = (size_t)nblk * S20_BS;
  for (unsigned blk = beg; blk < end; blk += nwaves)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    int se = 1, so = 1;
    for (unsigned r = 0; r < R; ++r)
    {
      const uint8_t * v = votes + ((size_t)opi * R + r) * nsite;
      se &= v[site0];
      so &= v[site0 + 1];
    }
    const int pe = pred_in[site0], po = pred_in[site0 + 1];
    if (__any((se != pe) | (so != po)))
    {
      const double fe = (se == pe) ? 1.0 : (se ? SCALE_FACTOR : 1.0 / SCALE_FACTOR);
      const double fo = (so == po) ? 1.0 : (so ? SCALE_FACTOR : 1.0 / SCALE_FACTOR);
      for (unsigned r = 0; r < R; ++r)
      {
        double * unit = op.parent + ((size_t)blk * R + r) * S61_UNIT;
        double2 t[S61_KS];
        s61_load_d(unit, lane, t);
#pragma unroll
        for (unsigned k = 0; k < S61_KS; ++k) { t[k].x *= fe; t[k].y *= fo; }
        s61_store_d(unit, lane, t);
      }
    }
    if (q == 0)
    {
      unsigned ce = se ? 1u : 0u, co = so ? 1u : 0u;
      if (op.scaler1) { ce += op.scaler1[site0]; co += op.scaler1[site0 + 1]; }
      if (op.scaler2) { ce += op.scaler2[site0]; co += op.scaler2[site0 + 1]; }
      op.parent_scaler[site0] = ce;
      op.parent_scaler[site0 + 1] = co;
      pred_out[site0] = (uint8_t)se;
      pred_out[site0 + 1] = (uint8_t)so;
    }
  }
}

// Rate-parallel launches (gridDim.z = R, used when a GPU has few blocks per wave):
// a workgroup handles ONE rate, writes its per-site "all entries small" vote to
// votes[(op*R + r)*Nalloc + site], and k_s61_scale_fixup combines the votes.
template <unsigned SC>      // the state count, 0 = a run-time value (S_rt, rows of the matrices Sp_rt apart)
__global__ __launch_bounds__(256, 2) void k_partials_s61v3(OpBatch batch, unsigned nblk, unsigned R,
                                                           unsigned lut_codes, uint8_t * votes, PredBatch preds,
                                                           unsigned rate_scalers, unsigned S_rt, unsigned Sp_rt)
{
  const unsigned S = SC ? SC : S_rt, Sp = SC ? S61_SP : Sp_rt;
  const bool rs = rate_scalers != 0;
  extern __shared__ double frag[];
  double * const frag2 = frag + S61_FRAGS;
  const OpDesc & op = batch.op[blockIdx.y];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const bool scaling = op.parent_scaler != nullptr && !rs;     // per-rate scalers are settled unit by unit
  const bool lut_lds = lut_codes * S <= S61_FRAGS;
  const bool tip1 = op.codes1 != nullptr, tip2 = op.codes2 != nullptr;
  const unsigned beg = (unsigned)(((unsigned long long)nblk * blockIdx.x) / gridDim.x);
  const unsigned end = (unsigned)(((unsigned long long)nblk * (blockIdx.x + 1)) / gridDim.x);
  const bool rate_parallel = gridDim.z > 1;
  // rate-parallel: units are stored with the scaling the LAST evaluation of this vector
  // decided; k_s61_scale_fixup corrects the sites where the new vote differs (rare)
  const uint8_t * pred = (rate_parallel && scaling) ? preds.in[blockIdx.y] : nullptr;
  const unsigned r_begin = rate_parallel ? blockIdx.z : 0, r_end = rate_parallel ? blockIdx.z + 1 : R;

  for (unsigned c0 = beg; c0 < end; c0 += S61_CHUNK)
  {
    const unsigned c1 = min(end, c0 + S61_CHUNK);
    const unsigned first = c0 + wave;
    const unsigned nb = first < c1 ? (c1 - first + 3) / 4 : 0;     // blocks first, first+4, ...
    unsigned small_e = ~0u, small_o = ~0u;                          // bit i: block i all-small so far

    for (unsigned r = r_begin; r < r_end; ++r)
    {
      __syncthreads();            // every wave is done reading the previous rate's fragments
      if (!tip1) s61_fill_frags_v3(frag, op.pmat1, r, S, Sp);
      else if (lut_lds) staged_copy<16>(frag, op.lut1 + (size_t)r * lut_codes * S, lut_codes * S);
      if (!tip2) s61_fill_frags_v3(frag2, op.pmat2, r, S, Sp);
      else if (lut_lds) staged_copy<16>(frag2, op.lut2 + (size_t)r * lut_codes * S, lut_codes * S);
      __syncthreads();
      if (nb == 0) continue;
      const double * l1 = lut_lds ? frag : op.lut1 + (size_t)r * lut_codes * S;
      const double * l2 = lut_lds ? frag2 : op.lut2 + (size_t)r * lut_codes * S;
      if (tip1 && tip2) s61_rate_tt(op, l1, l2, r, R, first, nb, lane, small_e, small_o, pred, rs, S);
      else if (!tip1 && !tip2) s61_rate_ii(op, frag, frag2, r, R, first, nb, lane, small_e, small_o, pred, rs);
      else if (tip1) s61_rate_ti(op, op.clv2, frag2, op.codes1, l1, r, R, first, nb, lane, small_e, small_o, pred, rs, S);
      else s61_rate_ti(op, op.clv1, frag, op.codes2, l2, r, R, first, nb, lane, small_e, small_o, pred, rs, S);
    }

    if (scaling && rate_parallel)
    {
      uint8_t * v = votes + ((size_t)blockIdx.y * R + r_begin) * ((size_t)nblk * S20_BS);
      for (unsigned i = 0; i < nb; ++i)
      {
        const int se = s20_and_q((int)((small_e >> i) & 1u)), so = s20_and_q((int)((small_o >> i) & 1u));
        if (q == 0)
        {
          const size_t site0 = (size_t)(first + 4 * i) * S20_BS + 2 * n;
          v[site0] = (uint8_t)se;
          v[site0 + 1] = (uint8_t)so;
        }
      }
    }
    else if (scaling)
    {
      for (unsigned i = 0; i < nb; ++i)
      {
        const unsigned blk = first + 4 * i;
        const int se = s20_and_q((int)((small_e >> i) & 1u)), so = s20_and_q((int)((small_o >> i) & 1u));
        if (__any(se | so))
        {
          const double fe = se ? SCALE_FACTOR : 1.0, fo = so ? SCALE_FACTOR : 1.0;
          for (unsigned r = 0; r < R; ++r)
          {
            double * unit = op.parent + ((size_t)blk * R + r) * S61_UNIT;
            double2 t[S61_KS];
            s61_load_d(unit, lane, t);
#pragma unroll
            for (unsigned k = 0; k < S61_KS; ++k) { t[k].x *= fe; t[k].y *= fo; }
            s61_store_d(unit, lane, t);
          }
        }
        if (q == 0)
        {
          const size_t site0 = (size_t)blk * S20_BS + 2 * n;
          unsigned ce = se ? 1u : 0u, co = so ? 1u : 0u;
          if (op.scaler1) { ce += op.scaler1[site0]; co += op.scaler1[site0 + 1]; }
          if (op.scaler2) { ce += op.scaler2[site0]; co += op.scaler2[site0 + 1]; }
          op.parent_scaler[site0] = ce;
          op.parent_scaler[site0 + 1] = co;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Cherries folded into their consumer.  A tip x tip operation ("cherry") does no matrix work: it
// looks two table rows up per site and streams one vector out, at HBM-write speed, while the
// matrix cores idle (1.4 of the 8.8 ms of a 50-taxon codon traversal).  Its only reader is the
// operation above it, which is bound by the matrix pipe -- so that operation computes the cherry
// itself: the B operand of its first matrix product is the product of the two tips' table rows,
// built in registers from LDS, and the same registers are stored to the cherry's vector (later
// partial traversals read it) while the MFMAs run.  The cherry's launch, its HBM-bound time and
// the re-read of its vector disappear.
//
// Scaling stays exact without a vote: whether a cherry site is scaled depends on its pair of tip
// codes only, so k_s61_cherry_scale tabulates the decision per code pair before the traversal
// and both the stored vector and the operand carry the right factor from the start.
//
// One 512-thread workgroup per CU (four tables: P fragments of the cherry's branch, the other
// child's fragments or table, the two tips' tables = 128 KiB of LDS), always rate-parallel.
// ---------------------------------------------------------------------------
constexpr unsigned S61_V4_OPS = 12;
constexpr unsigned S61_V4_WAVES = 8;

struct S61Batch
{
  OpDesc op[S61_V4_OPS];                 // child 1 is the folded cherry's vector when tt[i].parent != nullptr
  OpDesc tt[S61_V4_OPS];                 // the folded cherry: codes1/2, lut1/2, parent, scalers
  const uint8_t * ttscale[S61_V4_OPS];   // [codes][codes] per code pair: bit r = "every entry of rate r is small" (null: unscaled)
  const uint8_t * pred_in[S61_V4_OPS];   // predicted scaling of op[i] (see k_partials_s61v3)
};

struct CherryScaleBatch
{
  const double * lut1[32];
  const double * lut2[32];
  uint8_t * out[32];
};

// grid = (ceil(codes^2 / 1024), cherries), block = 1024, dynamic LDS = 2 x codes x 61 doubles: the two
// tables of a rate are staged with coalesced loads, a thread then owns one pair of codes
__global__ __launch_bounds__(1024) void k_s61_cherry_scale(CherryScaleBatch batch, unsigned lut_codes, unsigned R, unsigned S)
{
  extern __shared__ double tabs[];
  const unsigned len = lut_codes * S, pairs = lut_codes * lut_codes;
  double * ta = tabs, * tb = tabs + len;
  const double * l1 = batch.lut1[blockIdx.y], * l2 = batch.lut2[blockIdx.y];
  const unsigned pr = blockIdx.x * 1024u + threadIdx.x;
  const unsigned ca = pr < pairs ? pr / lut_codes : 0, cb = pr < pairs ? pr % lut_codes : 0;
  unsigned mask = 0u;                                   // bit r: every entry of rate r is small
  for (unsigned r = 0; r < R; ++r)
  {
    __syncthreads();
    staged_loop<4>(len, [=](unsigned e) { return make_double2(l1[(size_t)r * len + e], l2[(size_t)r * len + e]); },
                   [=](unsigned e, double2 v) { ta[e] = v.x; tb[e] = v.y; });
    __syncthreads();
    const double * ra = ta + ca * S, * rb = tb + cb * S;
    unsigned small = 1u;
    for (unsigned i = 0; i < S; ++i) small &= (ra[i] * rb[i] < SCALE_THRESHOLD) ? 1u : 0u;
    mask |= small << r;
  }
  if (pr < pairs) batch.out[blockIdx.y][pr] = (uint8_t)mask;
}

// k-steps 8 HALF .. 8 HALF + 7 of a cherry's unit: product of the two tips' rows with the cherry's
// scaling factor, stored to the cherry's vector and kept as B operand
template <unsigned HALF>
__device__ inline void s61_cherry_half(const double * ae, const double * ao, const double * be, const double * bo,
                                       unsigned q, double fe, double fo, double * unit, unsigned lane, double2 b[8], unsigned S)
{
#pragma unroll
  for (unsigned k = 0; k < 8; ++k)
  {
    const unsigned i = 4 * (HALF * 8 + k) + q;
    double2 v = make_double2(0.0, 0.0);
    if (i < S)
    {
      v.x = ae[i] * be[i];
      v.y = ao[i] * bo[i];
      v.x *= fe;
      v.y *= fo;
    }
    b[k] = v;
    *reinterpret_cast<double2 *>(unit + (HALF * 8 + k) * 128 + lane * 2) = v;
  }
}

// one rate of a wave's blocks: child 1 is a folded cherry, child 2 an inner vector (tip2 == false)
// or a coded tip (its table at lut2_r)
template <bool rs>
__device__ inline void s61_rate_cherry(const OpDesc & op, const OpDesc & tt, const uint8_t * ttscale, unsigned lut_codes,
                                       const double * frag1, const double * frag2, const double * lut2_r,
                                       const double * luta_r, const double * lutb_r, bool tip2,
                                       unsigned r, unsigned R, unsigned first, unsigned nb, unsigned lane,
                                       unsigned & small_e, unsigned & small_o, const uint8_t * pred, unsigned W, unsigned S)
{
  const unsigned all_rates = (1u << R) - 1u;
  const double2 * f1 = reinterpret_cast<const double2 *>(frag1);
  const double2 * f2 = reinterpret_cast<const double2 *>(frag2);
  const unsigned q = lane >> 4, n = lane & 15;
  double2 bA[8], bB[8], bC[8];
  if (!tip2) s61_issue_half<0>(op.clv2 + ((size_t)first * R + r) * S61_UNIT, lane, bA);
  size_t site0 = (size_t)first * S20_BS + 2 * n;
  unsigned cae = tt.codes1[site0], cao = tt.codes1[site0 + 1];
  unsigned cbe = tt.codes2[site0], cbo = tt.codes2[site0 + 1];
  for (unsigned i = 0; i < nb; ++i)
  {
    const unsigned blk = first + W * i;
    const unsigned blkn = first + W * (i + 1 < nb ? i + 1 : i);
    const size_t ub = ((size_t)blk * R + r) * S61_UNIT, ubn = ((size_t)blkn * R + r) * S61_UNIT;
    site0 = (size_t)blk * S20_BS + 2 * n;
    const size_t siten = (size_t)blkn * S20_BS + 2 * n;
    const unsigned nae = tt.codes1[siten], nao = tt.codes1[siten + 1];
    const unsigned nbe = tt.codes2[siten], nbo = tt.codes2[siten + 1];
    unsigned c2e = 0, c2o = 0;
    if (tip2) { c2e = op.codes2[site0]; c2o = op.codes2[site0 + 1]; }
    // the cherry's own scaling: a function of the code pair
    unsigned de = 0, dd = 0;
    if (ttscale)
    {
      // per-site scaling: all rates small; per-rate scalers: this rate small
      const unsigned me = ttscale[(size_t)cae * lut_codes + cbe], mo = ttscale[(size_t)cao * lut_codes + cbo];
      de = rs ? (me >> r) & 1u : (me == all_rates ? 1u : 0u);
      dd = rs ? (mo >> r) & 1u : (mo == all_rates ? 1u : 0u);
    }
    if (tt.parent_scaler && q == 0 && (rs || r == 0))
    {
      const size_t ie = rs ? site0 * R + r : site0, io = rs ? (site0 + 1) * R + r : site0 + 1;
      unsigned ce = de, co = dd;
      if (tt.scaler1) { ce += tt.scaler1[ie]; co += tt.scaler1[io]; }
      if (tt.scaler2) { ce += tt.scaler2[ie]; co += tt.scaler2[io]; }
      tt.parent_scaler[ie] = ce;
      tt.parent_scaler[io] = co;
    }
    const double tfe = de ? SCALE_FACTOR : 1.0, tfo = dd ? SCALE_FACTOR : 1.0;
    const double * ae = luta_r + cae * S, * ao = luta_r + cao * S;
    const double * be = lutb_r + cbe * S, * bo = lutb_r + cbo * S;

    v4d acc[S61_MT][2];
    double2 t1[S61_KS], t2[S61_KS];
    s61_acc_zero(acc);
    if (!tip2) s61_issue_half<1>(op.clv2 + ub, lane, bB);
    s61_cherry_half<0>(ae, ao, be, bo, q, tfe, tfo, tt.parent + ub, lane, bC, S);
    S61_SCHED_FENCE();
    s61_mfma_half<0>(bC, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_cherry_half<1>(ae, ao, be, bo, q, tfe, tfo, tt.parent + ub, lane, bC, S);
    S61_SCHED_FENCE();
    s61_mfma_half<1>(bC, f1, lane, acc);
    S61_SCHED_FENCE();
    s61_acc_to_t(acc, t1);
    if (!tip2)
    {
      s61_acc_zero(acc);
      s61_mfma_half<0>(bA, f2, lane, acc);
      S61_SCHED_FENCE();
      s61_issue_half<0>(op.clv2 + ubn, lane, bA);      // next block (the last one re-reads itself)
      S61_SCHED_FENCE();
      s61_mfma_half<1>(bB, f2, lane, acc);
      S61_SCHED_FENCE();
      s61_acc_to_t(acc, t2);
    }
    else s61_child_tip(lut2_r, c2e, c2o, q, t2, S);
    if (rs) s61_finish_unit_rs(op, blk, r, R, op.parent + ub, lane, t1, t2);
    else
    {
      double fe, fo;
      s61_pred_factors(pred, blk, lane, fe, fo);
      s61_finish_unit(op.parent + ub, lane, q, t1, t2, 1u << i, small_e, small_o, fe, fo);
    }
    cae = nae; cao = nao; cbe = nbe; cbo = nbo;
  }
}

// grid = (<= CUs / R, ops, R), block = 512, dynamic LDS = 4 x 32 KiB
template <bool rs, unsigned SC>   // rs: PLL_ATTRIB_RATE_SCALERS (a template parameter: as a run-time flag it cost the per-site form 54 spilled VGPRs); SC: see k_partials_s61v3
__global__ __launch_bounds__(64 * S61_V4_WAVES, 1) void k_partials_s61v4(S61Batch batch, unsigned nblk, unsigned R,
                                                                         unsigned lut_codes, uint8_t * votes,
                                                                         unsigned S_rt, unsigned Sp_rt)
{
  const unsigned S = SC ? SC : S_rt, Sp = SC ? S61_SP : Sp_rt;
  extern __shared__ double frag[];
  double * const frag2 = frag + S61_FRAGS, * const luta = frag + 2 * S61_FRAGS, * const lutb = frag + 3 * S61_FRAGS;
  const OpDesc & op = batch.op[blockIdx.y];
  const OpDesc & tt = batch.tt[blockIdx.y];
  const bool cherry = tt.parent != nullptr;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const bool scaling = op.parent_scaler != nullptr && !rs;      // per-rate scalers are settled unit by unit
  const bool tip1 = op.codes1 != nullptr, tip2 = op.codes2 != nullptr;
  const unsigned beg = (unsigned)(((unsigned long long)nblk * blockIdx.x) / gridDim.x);
  const unsigned end = (unsigned)(((unsigned long long)nblk * (blockIdx.x + 1)) / gridDim.x);
  const uint8_t * pred = scaling ? batch.pred_in[blockIdx.y] : nullptr;
  const unsigned r = blockIdx.z;
  const size_t lut_len = (size_t)lut_codes * S;

  if (!tip1) s61_fill_frags_v3(frag, op.pmat1, r, S, Sp);
  else staged_copy<8>(frag, op.lut1 + r * lut_len, (unsigned)lut_len);
  if (!tip2) s61_fill_frags_v3(frag2, op.pmat2, r, S, Sp);
  else staged_copy<8>(frag2, op.lut2 + r * lut_len, (unsigned)lut_len);
  if (cherry)
  {
    staged_copy<8>(luta, tt.lut1 + r * lut_len, (unsigned)lut_len);
    staged_copy<8>(lutb, tt.lut2 + r * lut_len, (unsigned)lut_len);
  }
  __syncthreads();

  constexpr unsigned W = S61_V4_WAVES;
  for (unsigned c0 = beg; c0 < end; c0 += S61_CHUNK)
  {
    const unsigned c1 = min(end, c0 + S61_CHUNK);
    const unsigned first = c0 + wave;
    const unsigned nb = first < c1 ? (c1 - first + W - 1) / W : 0;
    unsigned small_e = ~0u, small_o = ~0u;
    if (nb == 0) continue;
    if (cherry) s61_rate_cherry<rs>(op, tt, batch.ttscale[blockIdx.y], lut_codes, frag, frag2, frag2, luta, lutb, tip2,
                                    r, R, first, nb, lane, small_e, small_o, pred, W, S);
    else if (tip1 && tip2) s61_rate_tt(op, frag, frag2, r, R, first, nb, lane, small_e, small_o, pred, rs, S, W);
    else if (!tip1 && !tip2) s61_rate_ii(op, frag, frag2, r, R, first, nb, lane, small_e, small_o, pred, rs, W);
    else if (tip1) s61_rate_ti(op, op.clv2, frag2, op.codes1, frag, r, R, first, nb, lane, small_e, small_o, pred, rs, S, W);
    else s61_rate_ti(op, op.clv1, frag, op.codes2, frag2, r, R, first, nb, lane, small_e, small_o, pred, rs, S, W);

    if (scaling)
    {
      uint8_t * v = votes + ((size_t)blockIdx.y * R + r) * ((size_t)nblk * S20_BS);
      for (unsigned i = 0; i < nb; ++i)
      {
        const int se = s20_and_q((int)((small_e >> i) & 1u)), so = s20_and_q((int)((small_o >> i) & 1u));
        if (q == 0)
        {
          const size_t site0 = (size_t)(first + W * i) * S20_BS + 2 * n;
          v[site0] = (uint8_t)se;
          v[site0 + 1] = (uint8_t)so;
        }
      }
    }
  }
}

// grid = (blocks/4, ops), block = 256.  (Run by the last-arriving rate workgroup of a range inside the partials
// kernel instead -- one launch per operation -- it was twice as slow: the release / acquire fences around the
// arrival counter write back and invalidate the L2 of the XCD, which holds the units just written: SPR round at
// 25 k sites 0.93 -> 1.83 s, a 50-taxon traversal at 200 k sites 7.5 -> 10.8 ms.)
__global__ __launch_bounds__(256) void k_s61_scale_fixup(OpBatch batch, unsigned nblk, unsigned R,
                                                         const uint8_t * votes, PredBatch preds)
{
  const OpDesc & op = batch.op[blockIdx.y];
  if (!op.parent_scaler) return;
  const unsigned blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblk) return;
  s61_fixup_range(op, blockIdx.y, blk, blk + 1, 1, nblk, R, votes, preds.in[blockIdx.y], preds.out[blockIdx.y], threadIdx.x & 63);
}

// ---------------------------------------------------------------------------
// edge / root log-likelihood.  grid <= REDUCE_BLOCKS tiles-strided, block = 256
// dynamic LDS = S61_FRAGS doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_edge_lnl_s61(ModelView mv, ParamIdx fidx,
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
  const unsigned S = mv.S, Sp = mv.Sp;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned ntiles = (nblk + 3) / 4;
  double acc = 0.0;

  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
  {
    const unsigned blk = tile * 4 + wave;
    const bool live = blk < nblk;
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    SiteSide sd = load_site_side(rate_scalers ? nullptr : ps, rate_scalers ? nullptr : cs, weights, site0, N,
                                 q == 0 && blk < nblk);
    if (rate_scalers && live)
    {
      sd.cnt_e = rate_min_count(ps, cs, site0, R);
      sd.cnt_o = rate_min_count(ps, cs, site0 + 1, R);
    }
    unsigned cce = 0, cco = 0;
    unsigned long long pme = 0, pmo = 0;
    int inv_e = -1, inv_o = -1;
    if (live)
    {
      if (child.codes) { cce = child.codes[site0]; cco = child.codes[site0 + 1]; }
      if (parent.codes) { pme = tipmap[parent.codes[site0]]; pmo = tipmap[parent.codes[site0 + 1]]; }
      if (invariant)
      {
        inv_e = (site0 < N) ? invariant[site0] : -1;
        inv_o = (site0 + 1 < N) ? invariant[site0 + 1] : -1;
      }
    }
    double site_e = 0.0, site_o = 0.0, ie = 0.0, io = 0.0;
    for (unsigned r = 0; r < R; ++r)
    {
      __syncthreads();
      if (pmat && !child.codes) s61_fill_frags(frag, pmat, r, S, Sp);
      __syncthreads();
      if (!live) continue;
      const size_t ubase = ((size_t)blk * R + r) * S61_UNIT;
      const unsigned fi = fidx.v[r];
      const double * pi = mv.freqs(fi);
      double2 t[S61_KS], pv[S61_KS];
      if (!pmat)
      {
#pragma unroll
        for (unsigned k = 0; k < S61_KS; ++k) t[k] = make_double2(1.0, 1.0);
      }
      else if (child.codes) s61_child_tip(lut + (size_t)r * lut_codes * S, cce, cco, q, t, S);
      else s61_child_inner(child.clv + ubase, frag, lane, t);
      if (parent.codes) s61_tip_d(pme, pmo, q, pv, S);
      else s61_load_d(parent.clv + ubase, lane, pv);
      double le = 0.0, lo = 0.0;
#pragma unroll
      for (unsigned k = 0; k < S61_KS; ++k)
      {
        const unsigned i = 4 * k + q;
        const double f = (i < S) ? pi[i] : 0.0;
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
        if (inv_e >= 0) ie += w * pinv * pi[inv_e];
        if (inv_o >= 0) io += w * pinv * pi[inv_o];
      }
      else
      {
        site_e += w * le;
        site_o += w * lo;
      }
    }
    if (live && q == 0)
    {
      if (site0 < N)
      {
        const unsigned cnt = sd.cnt_e;
        const double l = site_loglh(site_e, cnt, ie);
        if (persite) persite[site0] = l;
        acc += l * (double)sd.w_e;
      }
      if (site0 + 1 < N)
      {
        const unsigned cnt = sd.cnt_o;
        const double l = site_loglh(site_o, cnt, io);
        if (persite) persite[site0 + 1] = l;
        acc += l * (double)sd.w_o;
      }
    }
  }
  const double tot = block_sum_256(acc, scratch);
  grid_reduce_finish1(tot, block_out, scratch);
}

// edge log-likelihood, inner child, R <= 4: the A fragments of ALL rates stay in LDS
// (R x 32 KiB), so a workgroup fills them once and every wave then streams its site
// blocks without a barrier; child operands are fetched half a unit ahead of the MFMAs
// (as in k_partials_s61v3), the parent unit of a rate during that rate's MFMAs.
// grid <= #CUs (one workgroup per CU, one wave per SIMD with the whole register file), dynamic LDS = (R * S61_FRAGS + R * 64) doubles.
// Block totals go to block_out[0 .. nreduce); slots beyond the grid are zeroed.
__global__ __launch_bounds__(256, 1) void k_edge_lnl_s61_r4(ModelView mv, ParamIdx fidx,
                                                            NodeRef parent, const double * child_clv,
                                                            const double * pmat,
                                                            const unsigned * ps, const unsigned * cs,
                                                            const unsigned * weights, const int * invariant,
                                                            const unsigned long long * tipmap,
                                                            unsigned N, unsigned nblk, unsigned R,
                                                            double * persite, ReduceOut block_out,
                                                            unsigned nreduce)
{
  extern __shared__ double frag[];
  __shared__ double scratch[4];
  double * const fq = frag + (size_t)R * S61_FRAGS;          // [R][64] frequencies, zero padded
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned beg = (unsigned)(((unsigned long long)nblk * blockIdx.x) / gridDim.x);
  const unsigned end = (unsigned)(((unsigned long long)nblk * (blockIdx.x + 1)) / gridDim.x);

  for (unsigned r = 0; r < R; ++r) s61_fill_frags_v3(frag + (size_t)r * S61_FRAGS, pmat, r, mv.S, mv.Sp);
  staged_loop<4>(R * 64, [=](unsigned e) { const bool in = (e & 63) < mv.S; const double x = mv.freqs(fidx.v[e >> 6])[in ? (e & 63) : 0]; return in ? x : 0.0; },
                 [=](unsigned e, double x) { fq[e] = x; });
  __syncthreads();

  double acc_lnl = 0.0;
  const unsigned first = beg + wave;
  const unsigned nb = first < end ? (end - first + 3) / 4 : 0;
  double2 bA[8], bB[8];
  if (nb) s61_issue_half<0>(child_clv + (size_t)first * R * S61_UNIT, lane, bA);

  for (unsigned i = 0; i < nb; ++i)
  {
    const unsigned blk = first + 4 * i;
    const unsigned blkn = first + 4 * (i + 1 < nb ? i + 1 : i);
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    const SiteSide sd = load_site_side(ps, cs, weights, site0, N, q == 0 && blk < nblk);
    unsigned long long pme = 0, pmo = 0;
    int inv_e = -1, inv_o = -1;
    if (parent.codes) { pme = tipmap[parent.codes[site0]]; pmo = tipmap[parent.codes[site0 + 1]]; }
    if (invariant)
    {
      inv_e = (site0 < N) ? invariant[site0] : -1;
      inv_o = (site0 + 1 < N) ? invariant[site0 + 1] : -1;
    }
    double site_e = 0.0, site_o = 0.0, ie = 0.0, io = 0.0;
    for (unsigned r = 0; r < R; ++r)
    {
      const size_t ub = ((size_t)blk * R + r) * S61_UNIT;
      const size_t ubn = (r + 1 < R) ? ub + S61_UNIT : (size_t)blkn * R * S61_UNIT;
      const double2 * fr = reinterpret_cast<const double2 *>(frag + (size_t)r * S61_FRAGS);
      v4d acc[S61_MT][2];
      double2 pv[S61_KS];
      s61_acc_zero(acc);
      s61_issue_half<1>(child_clv + ub, lane, bB);
      if (parent.codes) s61_tip_d(pme, pmo, q, pv, mv.S);
      else s61_load_d(parent.clv + ub, lane, pv);
      S61_SCHED_FENCE();
      s61_mfma_half<0>(bA, fr, lane, acc);
      S61_SCHED_FENCE();
      s61_issue_half<0>(child_clv + ubn, lane, bA);      // next rate / next block (the last re-reads)
      S61_SCHED_FENCE();
      s61_mfma_half<1>(bB, fr, lane, acc);
      S61_SCHED_FENCE();
      double le = 0.0, lo = 0.0;
#pragma unroll
      for (unsigned mt = 0; mt < S61_MT; ++mt)
#pragma unroll
        for (unsigned v = 0; v < 4; ++v)
        {
          const unsigned k = mt * 4 + v;
          const double f = fq[r * 64 + 4 * k + q];
          le += f * pv[k].x * acc[mt][0][v];
          lo += f * pv[k].y * acc[mt][1][v];
        }
      le = s20_sum_q(le);
      lo = s20_sum_q(lo);
      const unsigned fi = fidx.v[r];
      const double pinv = mv.pinv()[fi], w = mv.weights()[r];
      if (pinv > 0.0)
      {
        site_e += w * (1.0 - pinv) * le;
        site_o += w * (1.0 - pinv) * lo;
        if (inv_e >= 0) ie += w * pinv * fq[r * 64 + inv_e];
        if (inv_o >= 0) io += w * pinv * fq[r * 64 + inv_o];
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
        const double l = site_loglh(site_e, cnt, ie);
        if (persite) persite[site0] = l;
        acc_lnl += l * (double)sd.w_e;
      }
      if (site0 + 1 < N)
      {
        const unsigned cnt = sd.cnt_o;
        const double l = site_loglh(site_o, cnt, io);
        if (persite) persite[site0 + 1] = l;
        acc_lnl += l * (double)sd.w_o;
      }
    }
  }
  // 4 waves -> one total, fixed order
  acc_lnl = wave_sum(acc_lnl);
  if (lane == 0) scratch[wave] = acc_lnl;
  __syncthreads();
  const double tot = (threadIdx.x == 0) ? (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]) : 0.0;
  __syncthreads();
  // two-launch form: k_final_sum adds nreduce slots, the ones beyond this grid are zero
  if (!block_out.fused && threadIdx.x == 0)
    for (unsigned b = blockIdx.x + gridDim.x; b < nreduce; b += gridDim.x) block_out.block_out[b] = 0.0;
  grid_reduce_finish1(tot, block_out, scratch);
}

// sumtable preparation: Lm[r][k][i] = pi_i V[i][k], Rm[r][k][j] = V^-1[k][j] in
// [r][S][Sp] row-major form, plus tip lookup tables [r][code][S]
__global__ __launch_bounds__(256) void k_sumtable_prep_s61(ModelView mv, ParamIdx params,
                                                           const unsigned long long * tipmap,
                                                           unsigned lut_codes, bool want_lut,
                                                           double * Lm, double * Rm,
                                                           double * lutL, double * lutR)
{
  const unsigned r = blockIdx.x, pi_ = params.v[r];
  const unsigned S = mv.S, Sp = mv.Sp;
  const double * pi = mv.freqs(pi_), * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_);
  double * L = Lm + (size_t)r * S * Sp, * Rr = Rm + (size_t)r * S * Sp;
  staged_loop<8>(S * Sp, [=](unsigned e)
  {
    const unsigned k = e / Sp, i = e % Sp, ic = i < S ? i : 0;
    const double a = pi[ic] * V[ic * Sp + k], b = Vi[k * Sp + ic];
    return (i < S) ? make_double2(a, b) : make_double2(0.0, 0.0);
  }, [=](unsigned e, double2 v) { L[e] = v.x; Rr[e] = v.y; });
  if (!want_lut) return;
  for (unsigned e = threadIdx.x; e < lut_codes * S; e += blockDim.x)
  {
    const unsigned c = e / S, k = e % S;
    const unsigned long long mask = tipmap[c];
    double a = 0.0, b = 0.0;
    if (mask && !(mask & (mask - 1)))                 // one state (most codes): no walk over the alphabet
    {
      const unsigned i = (unsigned)__ffsll((long long)mask) - 1;
      a += pi[i] * V[i * Sp + k];
      b += Vi[k * Sp + i];
    }
    else
      for (unsigned i = 0; i < S; ++i)
        if ((mask >> i) & 1ULL) { a += pi[i] * V[i * Sp + k]; b += Vi[k * Sp + i]; }
    lutL[((size_t)r * lut_codes + c) * S + k] = a;
    lutR[((size_t)r * lut_codes + c) * S + k] = b;
  }
}

// derivatives of -lnL: k_derivatives_mfma<16, 61> (kernels_s20.hpp)

// --- launchers -------------------------------------------------------------

// Rate-parallel launches: one workgroup per (range of site blocks, rate), the scaling votes of the R
// workgroups of a site meet in a second kernel.  Introduced for slices with few blocks per wave
// (it balances the matrix pipes and quarters the critical path), it is the faster form at every
// size: 1 M sites, 50 taxa: 40.0 ms against 43.4 ms with the rates walked inside a workgroup
// (35.7 ms with the cherries folded, which needs this form).  PLLHIP_S61_RATEPAR=0 / 1 forces either.
static bool s61_rate_parallel(const Engine * e)
{
  static const int env_rp = getenv("PLLHIP_S61_RATEPAR") ? atoi(getenv("PLLHIP_S61_RATEPAR")) : -1;
  return e->R > 1 && (env_rp >= 0 ? env_rp != 0 : true);
}

// cherries can be folded into their consumers (k_partials_s61v4)
static bool s61_cherries_supported(const Engine * e)
{
  static const int env = getenv("PLLHIP_S61_CHERRIES") ? atoi(getenv("PLLHIP_S61_CHERRIES")) : 1;
  // (per-rate scalers are carried as well: C5 --rate-scalers 8.93 -> 7.86 ms)
  return env && (e->coded_tips || e->shadow_codes) && s61_rate_parallel(e) && e->lut_codes * e->S <= S61_FRAGS && e->R <= 8;
}

// last scaling decisions per parent vector, double-buffered (the fix-up kernel reads the
// old one in all its rate slices while one of them writes the new one)
static int s61_prepare_preds(Engine * e, const OpBatch & batch, unsigned nops, PredBatch & preds)
{
  if (!e->d_s61_votes)
    PLLHIP_TRY(hipMalloc((void **)&e->d_s61_votes, (size_t)MAX_OPS_PER_LAUNCH * e->R * e->nblk * S20_BS));
  if (e->s61_pred.empty()) e->s61_pred.resize(3 * (size_t)e->nodes);
  const size_t bytes = (size_t)e->nblk * S20_BS;
  for (unsigned i = 0; i < nops; ++i)
  {
    const OpDesc & d = batch.op[i];
    if (!d.parent_scaler) continue;
    const unsigned long long lo = std::min(d.child1_index, d.child2_index), hi = std::max(d.child1_index, d.child2_index);
    const unsigned long long key = (hi << 32) | lo;
    Engine::PredSlot * slot = nullptr, * victim = &e->s61_pred[3 * (size_t)d.parent_index], * latest = nullptr;
    for (int w = 0; w < 3; ++w)
    {
      Engine::PredSlot & c = e->s61_pred[3 * (size_t)d.parent_index + w];
      if (c.key == key) { slot = &c; break; }
      if (c.used < victim->used) victim = &c;
      if (c.used && c.buf[c.cur] && (!latest || c.used > latest->used)) latest = &c;
    }
    const uint8_t * from = nullptr;
    if (!slot)
    {
      // new orientation (or topology).  Start from what the vector's last evaluation decided, whatever its
      // children were: in an SPR round the scratch vectors near the root are re-evaluated over subtrees that
      // differ by the pruned part only, and about half of their sites scale -- "no site scales" made the
      // fix-up kernel rewrite most blocks there.  (The launch reads that slot's buffer in place; first
      // evaluation ever: no site scales.)
      from = latest ? latest->buf[latest->cur] : nullptr;
      slot = victim;
      slot->key = key;
      for (int w = 0; w < 2; ++w)
        if (!slot->buf[w]) PLLHIP_TRY(hipMalloc((void **)&slot->buf[w], bytes));
      if (!from)
      {
        slot->cur = 0;
        PLLHIP_TRY(hipMemsetAsync(slot->buf[0], 0, bytes, e->stream));
      }
      else if (from == slot->buf[0] || from == slot->buf[1]) from = nullptr;      // the victim's own last decision
      else slot->cur = 1;                       // written into buf[0] below
    }
    slot->used = ++e->s61_pred_clock;
    preds.in[i] = from ? from : slot->buf[slot->cur];
    preds.out[i] = slot->buf[slot->cur ^ 1u];
    slot->cur ^= 1u;
  }
  return PLL_SUCCESS;
}

static int launch_partials_s61(Engine * e, const OpBatch & batch, unsigned nops)
{
  const size_t lds = sizeof(double) * 2 * S61_FRAGS;
  const unsigned slots = e->cu_count * 2u;
  const bool rate_parallel = s61_rate_parallel(e);
  bool scaling = false;
  for (unsigned i = 0; i < nops; ++i) scaling |= batch.op[i].parent_scaler != nullptr;
  if (e->rate_scalers) scaling = false;      // per-rate scalers: no votes, no predictions, no fix-up kernel
  PredBatch preds;
  memset(&preds, 0, sizeof(preds));
  if (rate_parallel && scaling && !s61_prepare_preds(e, batch, nops, preds)) return PLL_FAILURE;
  static const int env_mul = getenv("PLLHIP_S61_GXMUL") ? atoi(getenv("PLLHIP_S61_GXMUL")) : 1;
  const unsigned per = (rate_parallel ? std::max(1u, slots / e->R) : slots) * (unsigned)std::max(1, env_mul);
  const unsigned gx = std::max(1u, std::min((e->nblk + 3) / 4, per));
  if (e->S == S61_S)
    hipLaunchKernelGGL(k_partials_s61v3<S61_S>, dim3(gx, nops, rate_parallel ? e->R : 1u), dim3(256), lds, e->stream,
                       batch, e->nblk, e->R, e->lut_codes, rate_parallel ? e->d_s61_votes : (uint8_t *)nullptr, preds,
                       e->rate_scalers ? 1u : 0u, e->S, e->Sp);
  else
    hipLaunchKernelGGL(k_partials_s61v3<0>, dim3(gx, nops, rate_parallel ? e->R : 1u), dim3(256), lds, e->stream,
                       batch, e->nblk, e->R, e->lut_codes, rate_parallel ? e->d_s61_votes : (uint8_t *)nullptr, preds,
                       e->rate_scalers ? 1u : 0u, e->S, e->Sp);
  PLLHIP_TRY(hipGetLastError());
  if (rate_parallel && scaling)
  {
    hipLaunchKernelGGL(k_s61_scale_fixup, dim3((e->nblk + 3) / 4, nops), dim3(256), 0, e->stream,
                       batch, e->nblk, e->R, (const uint8_t *)e->d_s61_votes, preds);
    PLLHIP_TRY(hipGetLastError());
  }
  return PLL_SUCCESS;
}

// a launch of the cherry-folding kernel: batch.op[i] with the cherry cherries.op[i] folded in
// (cherries.op[i].parent == nullptr: none), ttscale[i] its per-code-pair scaling table
static int launch_partials_s61_cherries(Engine * e, const OpBatch & batch, const OpBatch & cherries,
                                        const uint8_t * const * ttscale, unsigned nops)
{
  if (nops > S61_V4_OPS) { set_error(PLL_ERROR_PARAM_INVALID, "cherry launch of %u operations", nops); return PLL_FAILURE; }
  const size_t lds = sizeof(double) * 4 * S61_FRAGS;
  static bool attr_set_dev[64] = {false};
  bool & attr_set = attr_set_dev[e->device & 63];
  if (!attr_set)
  {
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partials_s61v4<false, S61_S>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partials_s61v4<true, S61_S>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partials_s61v4<false, 0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partials_s61v4<true, 0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  bool scaling = false;
  for (unsigned i = 0; i < nops; ++i) scaling |= batch.op[i].parent_scaler != nullptr;
  if (e->rate_scalers) scaling = false;      // per-rate scalers: no votes, no predictions, no fix-up kernel
  PredBatch preds;
  memset(&preds, 0, sizeof(preds));
  if (scaling && !s61_prepare_preds(e, batch, nops, preds)) return PLL_FAILURE;
  S61Batch sb;
  memset(&sb, 0, sizeof(sb));
  for (unsigned i = 0; i < nops; ++i)
  {
    sb.op[i] = batch.op[i];
    sb.tt[i] = cherries.op[i];
    sb.ttscale[i] = ttscale[i];
    sb.pred_in[i] = preds.in[i];
  }
  static const int env_per = getenv("PLLHIP_S61_V4_PER") ? atoi(getenv("PLLHIP_S61_V4_PER")) : 0;
  // workgroups per (operation, rate): at most a quarter of the CUs each, and about three workgroups per CU
  // over the whole launch -- fewer, longer ranges per table fill when a launch has many operations
  // (25 k / 50 k / 100 k / 200 k sites: 1.75 -> 1.68, 2.58 -> 2.52, 4.27 -> 4.23, 7.55 -> 7.57 ms per step)
  static const int env_fill = getenv("PLLHIP_S61_V4_FILL") ? atoi(getenv("PLLHIP_S61_V4_FILL")) : 3;
  unsigned per = env_per > 0 ? (unsigned)env_per : std::max(1u, e->cu_count / e->R);
  if (env_fill > 0) per = std::min(per, std::max(1u, (unsigned)env_fill * e->cu_count / (nops * e->R)));
  const unsigned gx = std::max(1u, std::min((e->nblk + S61_V4_WAVES - 1) / S61_V4_WAVES, per));
#define PLLHIP_CALL(RS, SCV) \
  hipLaunchKernelGGL((k_partials_s61v4<RS, SCV>), dim3(gx, nops, e->R), dim3(64 * S61_V4_WAVES), lds, e->stream, \
                     sb, e->nblk, e->R, e->lut_codes, e->d_s61_votes, e->S, e->Sp)
  if (e->S == S61_S) { if (e->rate_scalers) PLLHIP_CALL(true, S61_S); else PLLHIP_CALL(false, S61_S); }
  else { if (e->rate_scalers) PLLHIP_CALL(true, 0); else PLLHIP_CALL(false, 0); }
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  if (scaling)
  {
    hipLaunchKernelGGL(k_s61_scale_fixup, dim3((e->nblk + 3) / 4, nops), dim3(256), 0, e->stream,
                       batch, e->nblk, e->R, (const uint8_t *)e->d_s61_votes, preds);
    PLLHIP_TRY(hipGetLastError());
  }
  return PLL_SUCCESS;
}

// scaling tables of `count` cherries (CherryScaleBatch): out[i] = [codes][codes]
static int launch_cherry_scale_s61(Engine * e, const CherryScaleBatch & batch, unsigned count)
{
  const size_t lds = sizeof(double) * 2 * e->lut_codes * e->S;      // <= 64 KiB (s61_cherries_supported)
  hipLaunchKernelGGL(k_s61_cherry_scale, dim3((e->lut_codes * e->lut_codes + 1023u) / 1024u, count), dim3(1024), lds, e->stream,
                     batch, e->lut_codes, e->R, e->S);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_edge_lnl_s61(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                               const NodeRef & parent, const NodeRef & child,
                               const double * pm, const double * lut,
                               const unsigned * ps, const unsigned * cs,
                               double * persite, unsigned nblocks)
{
  static const int env_r4 = getenv("PLLHIP_S61_LNL_R4") ? atoi(getenv("PLLHIP_S61_LNL_R4")) : 1;
  if (env_r4 && pm && !child.codes && e->R <= 4 && e->nblk && !e->rate_scalers)
  {
    const size_t lds = sizeof(double) * ((size_t)e->R * S61_FRAGS + (size_t)e->R * 64);
    static bool attr_set_dev[64] = {false};        // per device: one process may drive several GPUs
    bool & attr_set = attr_set_dev[e->device & 63];
    if (!attr_set)
    {
      PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_lnl_s61_r4),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (S61_FRAGS + 64) * 8));
      attr_set = true;
    }
    const unsigned gx = std::max(1u, std::min({nblocks, e->cu_count, (e->nblk + 3) / 4}));
    hipLaunchKernelGGL(k_edge_lnl_s61_r4, dim3(gx), dim3(256), lds, e->stream,
                       mv, fidx, parent, child.clv, pm, ps, cs,
                       e->d_weights, e->d_invariant, e->d_tipmap, e->N, e->nblk, e->R,
                       persite, reduce_out(e), nblocks);
    PLLHIP_TRY(hipGetLastError());
    return PLL_SUCCESS;
  }
  const size_t lds = sizeof(double) * S61_FRAGS;
  hipLaunchKernelGGL(k_edge_lnl_s61, dim3(nblocks), dim3(256), lds, e->stream,
                     mv, fidx, parent, child, pm, lut, e->lut_codes, ps, cs,
                     e->d_weights, e->d_invariant, e->d_tipmap, e->N, e->nblk, e->R,
                     persite, reduce_out(e), e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_sumtable_s61(Engine * e, const ModelView & mv, const ParamIdx & params,
                               const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  const size_t mats = (size_t)e->R * e->S * e->Sp, luts = (size_t)e->R * std::max(1u, e->lut_codes) * e->S;
  if (!e->d_sum_scratch)
  {
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&e->d_sum_scratch),
                               sizeof(double) * 2 * (mats + (size_t)e->R * PLL_ASCII_SIZE * e->S));
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
    hipLaunchKernelGGL(k_sumtable_prep_s61, dim3(e->R), dim3(256), 0, e->stream,
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
  return launch_partials_s61(e, batch, 1);
}

static int launch_derivatives_s61(Engine * e, const ModelView & mv, const ParamIdx & params,
                                  const TrialLengths & tl, unsigned count,
                                  const double * d_sum, const unsigned * ps, const unsigned * cs,
                                  unsigned nblocks)
{
  const size_t lds = sizeof(double) * e->R * S61_KS * 64;
  if (lds > 160 * 1024 - 512)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "derivatives: %u rate categories exceed the LDS operand tables", e->R);
    return PLL_FAILURE;
  }
  if (lds > 64 * 1024)
  {
    static bool attr_set_dev[64] = {false};        // per device: one process may drive several GPUs
    bool & attr_set = attr_set_dev[e->device & 63];
    if (!attr_set)
    {
      PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_derivatives_mfma<S61_KS, S61_S>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
      PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_derivatives_mfma<S61_KS, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
      attr_set = true;
    }
  }
  if (e->S == S61_S)
    hipLaunchKernelGGL((k_derivatives_mfma<S61_KS, S61_S>), dim3(nblocks), dim3(256), lds, e->stream,
                       mv, params, tl, count, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->nblk, e->R, reduce_out(e),
                       e->rate_scalers ? 1u : 0u);
  else
    hipLaunchKernelGGL((k_derivatives_mfma<S61_KS, 0>), dim3(nblocks), dim3(256), lds, e->stream,
                       mv, params, tl, count, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->nblk, e->R, reduce_out(e),
                       e->rate_scalers ? 1u : 0u);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

} // namespace pllhip
