// kernels_s16.hpp -- 2 .. 32 states on the fp64 matrix cores: binary, genotype (10 / 16
// states, src/util/models_gt.c), multistate alphabets (up to 64 states, src/util/models_mult.c:92-97; this
// family takes them up to 32, the 61-state family from 33), and 4 / 20 states with a rate count their own
// families do not take.  Also the family that carries PLL_ATTRIB_RATE_SCALERS for these alphabets.
//
// Same blocked device layout and lane mapping as the 20- and 61-state families
// (kernels_s20.hpp): clv[site_block][rate][state row][32 sites]; a (block, rate) unit is a
// (4 KS) x 32 fp64 matrix with KS = ceil(S / 4) k-steps (rows >= S are zero); lane
// l = 16 q + n holds sites 2n, 2n + 1 and slot k of a lane is row 4 k + q -- for the MFMA B
// operand (k-step k) and the MFMA D result (register k) alike, so every CLV access is a fully
// coalesced 1 KiB wave instruction and what one operation stores is what the next one loads.
// MT = ceil(KS / 4) 16-row M tiles cover all states (one up to 16 states, two up to 32): MT x KS MFMAs per
// child, unit and site parity; slot k of a lane is D register k % 4 of M tile k / 4.  The kernels
// are HBM-bound at every S (S = 16: 512 MACs per 128 B loaded).
//
// Scaling: per site (all R x S entries below 2^-256: store unscaled, rare fix-up of the units
// just written) or per (site, rate) with PLL_ATTRIB_RATE_SCALERS (the vote covers one unit:
// decided and applied before the unit is stored; scaler[n * R + r]).
#pragma once

#include "kernels_common.hpp"
#include "kernels_s20.hpp"
#include "engine.h"

namespace pllhip {

// doubles of LDS a tip lookup table may take per child (28 states x 36 codes x 4 rates, 32 x 36 x 4: beyond it the rows
// are gathered from memory -- which cost the 28- and 32-state alphabets, half of whose operands are tips, a fifth of
// their rate while the limit was 3072)
constexpr unsigned S16_LUT_LDS = 4608;
constexpr int S16_MAX_LDS_BYTES = 160 * 1024 - 512;   // dynamic LDS a kernel of the family may ask for (it has a few static words)

constexpr unsigned s16_mt(unsigned KS) { return (KS + 3) / 4; }            // 16-row M tiles
constexpr unsigned s16_fr(unsigned KS) { return s16_mt(KS) * KS * 64; }     // fragment doubles per (child, rate)

// A fragments of a matrix set [R][S][Sp] (row-major) into LDS:
//   frag[((r * MT + mt) * KS + ks) * 64 + lane] = M[r][16 mt + (lane & 15)][4 ks + (lane >> 4)]   (0 beyond S)
template <unsigned KS>
__device__ inline void s16_fill_frags(double * frag, const double * mats, unsigned R, unsigned S, unsigned Sp)
{
  constexpr unsigned MT = s16_mt(KS);
  // (staged_loop, kernels_common.hpp: the loads of a thread go out together)
  staged_loop<8>(R * MT * KS * 64, [=](unsigned e)
  {
    const unsigned lane = e & 63, f = e >> 6, ks = f % KS, mt = (f / KS) % MT, r = f / (KS * MT);
    const unsigned i = 16 * mt + (lane & 15), j = 4 * ks + (lane >> 4);
    const bool in = i < S && j < S;
    const double x = mats[in ? ((size_t)r * S + i) * Sp + j : 0];
    return in ? x : 0.0;
  }, [=](unsigned e, double x) { frag[e] = x; });
}

// child term from a B operand in registers (b[ks] = rows 4 ks + q of the child vector), D layout out:
// t[v] = {even site, odd site} of row 4 v + q
template <unsigned KS>
__device__ inline void s16_child_regs(const double2 b[KS], const double * frag_r, unsigned lane, double2 t[KS])
{
  // Full 16-row tiles (four row groups) on v_mfma_f64_16x16x4_f64; the row groups of a PARTIAL last tile -- rows
  // 4 g .. 4 g + 3 each -- on v_mfma_f64_4x4x4_4b_f64 (kernels_s20.hpp, mfma_f64_tail): lane (q, n) reads
  // M[4 g + (n & 3)][4 ks + q], i.e. lane 16 q + 4 (g & 3) + (n & 3) of the tile's fragment, and gets row 4 g + q of the
  // product: slot g of the D layout.  Same inner sums, bit-identical results, 8 ns instead of 48 .. 65 ns per instruction
  // and no padded rows: 2 states 2.26 -> 2.07 ms per traversal (1 M sites, 50 taxa), 7 states 1.81 -> 1.72, 17 states
  // 4.70 -> 4.28, 24 states 6.42 -> 5.94 (500 k sites).  (Every group on the short instruction was measured as well:
  // 3 - 5 % slower at 16 and 32 states, where no row is padding: four times the LDS operand reads.)
  constexpr unsigned MTF = KS / 4, REM = KS % 4;          // full tiles, row groups of the partial one
  v4d acc_e[MTF ? MTF : 1], acc_o[MTF ? MTF : 1];
  double tail_e[REM ? REM : 1], tail_o[REM ? REM : 1];
#pragma unroll
  for (unsigned mt = 0; mt < MTF; ++mt) { acc_e[mt] = v4d{0, 0, 0, 0}; acc_o[mt] = v4d{0, 0, 0, 0}; }
#pragma unroll
  for (unsigned g = 0; g < REM; ++g) tail_e[g] = tail_o[g] = 0.0;
  const double * mine = frag_r + (MTF * KS) * 64 + (lane & 48u) + (lane & 3u);
#pragma unroll
  for (unsigned ks = 0; ks < KS; ++ks)
  {
#pragma unroll
    for (unsigned mt = 0; mt < MTF; ++mt)
    {
      const double f = frag_r[(mt * KS + ks) * 64 + lane];
      acc_e[mt] = mfma_f64(f, b[ks].x, acc_e[mt]);
      acc_o[mt] = mfma_f64(f, b[ks].y, acc_o[mt]);
    }
#pragma unroll
    for (unsigned g = 0; g < REM; ++g)
    {
      const double f = mine[ks * 64 + 4 * g];
      tail_e[g] = mfma_f64_tail(f, b[ks].x, tail_e[g]);
      tail_o[g] = mfma_f64_tail(f, b[ks].y, tail_o[g]);
    }
  }
#pragma unroll
  for (unsigned v = 0; v < KS; ++v)
    t[v] = (v < 4 * MTF) ? make_double2(acc_e[v / 4][v % 4], acc_o[v / 4][v % 4])
                         : make_double2(tail_e[(v - 4 * MTF) % (REM ? REM : 1)], tail_o[(v - 4 * MTF) % (REM ? REM : 1)]);
}

// child term in D layout from a unit in memory
template <unsigned KS>
__device__ inline void s16_child_inner(const double * unit, const double * frag_r, unsigned lane, double2 t[KS],
                                       bool nt = false)
{
  double2 b[KS];
  typedef double nt_v2d __attribute__((ext_vector_type(2)));
#pragma unroll
  for (unsigned ks = 0; ks < KS; ++ks)
  {
    if (nt)
    {
      const nt_v2d w = __builtin_nontemporal_load(reinterpret_cast<const nt_v2d *>(unit + ks * 128 + lane * 2));
      b[ks] = make_double2(w.x, w.y);
    }
    else b[ks] = *reinterpret_cast<const double2 *>(unit + ks * 128 + lane * 2);
  }
  s16_child_regs<KS>(b, frag_r, lane, t);
}

// the B operand of a unit in memory
template <unsigned KS>
__device__ inline void s16_load_b(const double * unit, unsigned lane, double2 b[KS], bool nt)
{
  typedef double nt_v2d __attribute__((ext_vector_type(2)));
#pragma unroll
  for (unsigned ks = 0; ks < KS; ++ks)
  {
    if (nt)
    {
      const nt_v2d w = __builtin_nontemporal_load(reinterpret_cast<const nt_v2d *>(unit + ks * 128 + lane * 2));
      b[ks] = make_double2(w.x, w.y);
    }
    else b[ks] = *reinterpret_cast<const double2 *>(unit + ks * 128 + lane * 2);
  }
}

// lookup table rows [code][S]
template <unsigned KS>
__device__ inline void s16_child_tip(const double * lut_r, unsigned code_e, unsigned code_o, unsigned q,
                                     unsigned S, double2 t[KS])
{
  const double * le = lut_r + code_e * S, * lo = lut_r + code_o * S;
#pragma unroll
  for (unsigned v = 0; v < KS; ++v)
  {
    const unsigned i = 4 * v + q;
    t[v] = (i < S) ? make_double2(le[i], lo[i]) : make_double2(0.0, 0.0);
  }
}

template <unsigned KS>
__device__ inline void s16_tip_d(unsigned long long mask_e, unsigned long long mask_o, unsigned q, unsigned S,
                                 double2 t[KS])
{
#pragma unroll
  for (unsigned v = 0; v < KS; ++v)
  {
    const unsigned i = 4 * v + q;
    t[v] = (i < S) ? make_double2((double)((mask_e >> i) & 1ULL), (double)((mask_o >> i) & 1ULL))
                   : make_double2(0.0, 0.0);
  }
}

template <unsigned KS>
__device__ inline void s16_load_d(const double * unit, unsigned lane, double2 t[KS])
{
#pragma unroll
  for (unsigned v = 0; v < KS; ++v) t[v] = *reinterpret_cast<const double2 *>(unit + v * 128 + lane * 2);
}

// (stores that do not allocate in the caches: a vector is written once and read, if at all, by a later chain)
template <unsigned KS>
__device__ inline void s16_store_d_nt(double * unit, unsigned lane, const double2 t[KS])
{
  typedef double nt_v2d __attribute__((ext_vector_type(2)));
#pragma unroll
  for (unsigned v = 0; v < KS; ++v)
  {
    nt_v2d w; w.x = t[v].x; w.y = t[v].y;
    __builtin_nontemporal_store(w, reinterpret_cast<nt_v2d *>(unit + v * 128 + lane * 2));
  }
}

template <unsigned KS>
__device__ inline void s16_store_d(double * unit, unsigned lane, const double2 t[KS])
{
#pragma unroll
  for (unsigned v = 0; v < KS; ++v) *reinterpret_cast<double2 *>(unit + v * 128 + lane * 2) = t[v];
}

// ---------------------------------------------------------------------------
// partials (also the sumtable, with eigen-basis matrices in place of the P-matrices)
// grid = (gx, ops), block = 256 (4 independent waves); dynamic LDS = 2 tables:
// per child R * KS * 64 doubles of A fragments, or its tip lookup table when that fits
// ---------------------------------------------------------------------------
template <unsigned KS>
// nt: vectors are read and written past the caches (a traversal: written once, read -- if at all -- by a later launch;
// not the sumtable, which the derivative scans read next)
__global__ __launch_bounds__(256, 2) void k_partials_s16(OpBatch batch, unsigned nblk, unsigned R, unsigned S,
                                                      unsigned Sp, unsigned lut_codes, unsigned table,
                                                      unsigned rate_scalers, unsigned nt)
{
  extern __shared__ double lds[];
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const OpDesc & op = batch.op[blockIdx.y];
  double * const tab1 = lds, * const tab2 = lds + table;
  const bool lut_lds = R * lut_codes * S <= table;
  if (!op.codes1) s16_fill_frags<KS>(tab1, op.pmat1, R, S, Sp);
  else if (lut_lds)
    staged_copy<8>(tab1, op.lut1, R * lut_codes * S);
  if (!op.codes2) s16_fill_frags<KS>(tab2, op.pmat2, R, S, Sp);
  else if (lut_lds)
    staged_copy<8>(tab2, op.lut2, R * lut_codes * S);
  __syncthreads();
  const double * l1 = lut_lds ? tab1 : op.lut1, * l2 = lut_lds ? tab2 : op.lut2;

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
    // From 17 states (two row tiles: 2 x KS MFMAs per child and site parity, 3 us of matrix time per unit and wave at 32
    // states) the operands of the next rate are requested before the products of this one are started: with two waves
    // per SIMD -- the tables of 32 states fill the LDS of a CU with two workgroups -- nothing else keeps loads in flight
    // while both of them multiply
    constexpr bool AHEAD = KS >= 5;
    double2 b1[AHEAD ? KS : 1], b2[AHEAD ? KS : 1];
    if (AHEAD)
    {
      if (!op.codes1) s16_load_b<KS>(op.clv1 + (size_t)blk * R * UNIT, lane, b1, nt != 0);
      if (!op.codes2) s16_load_b<KS>(op.clv2 + (size_t)blk * R * UNIT, lane, b2, nt != 0);
    }
    for (unsigned r = 0; r < R; ++r)
    {
      const size_t ubase = ((size_t)blk * R + r) * UNIT;
      double2 t1[KS], t2[KS];
      if (AHEAD)
      {
        double2 n1[KS], n2[KS];
        const bool more = r + 1 < R;
        if (more && !op.codes1) s16_load_b<KS>(op.clv1 + ubase + UNIT, lane, n1, nt != 0);
        if (more && !op.codes2) s16_load_b<KS>(op.clv2 + ubase + UNIT, lane, n2, nt != 0);
        if (!op.codes1) s16_child_regs<KS>(b1, tab1 + r * s16_fr(KS), lane, t1);
        else s16_child_tip<KS>(l1 + (size_t)r * lut_codes * S, c1e, c1o, q, S, t1);
        if (!op.codes2) s16_child_regs<KS>(b2, tab2 + r * s16_fr(KS), lane, t2);
        else s16_child_tip<KS>(l2 + (size_t)r * lut_codes * S, c2e, c2o, q, S, t2);
        if (more)
        {
#pragma unroll
          for (unsigned v = 0; v < KS; ++v) { b1[v] = n1[v]; b2[v] = n2[v]; }
        }
      }
      else
      {
        if (!op.codes1) s16_child_inner<KS>(op.clv1 + ubase, tab1 + r * s16_fr(KS), lane, t1, nt != 0);
        else s16_child_tip<KS>(l1 + (size_t)r * lut_codes * S, c1e, c1o, q, S, t1);
        if (!op.codes2) s16_child_inner<KS>(op.clv2 + ubase, tab2 + r * s16_fr(KS), lane, t2, nt != 0);
        else s16_child_tip<KS>(l2 + (size_t)r * lut_codes * S, c2e, c2o, q, S, t2);
      }
      int re = 1, ro = 1;
#pragma unroll
      for (unsigned v = 0; v < KS; ++v)
      {
        t1[v].x *= t2[v].x;
        t1[v].y *= t2[v].y;
        re &= (t1[v].x < SCALE_THRESHOLD);     // rows >= S are zero: they never veto
        ro &= (t1[v].y < SCALE_THRESHOLD);
      }
      if (scaling && rate_scalers)
      {
        // the vote covers this unit only: decide, scale, store, count
        re = s20_and_q(re);
        ro = s20_and_q(ro);
        const double fe = re ? SCALE_FACTOR : 1.0, fo = ro ? SCALE_FACTOR : 1.0;
#pragma unroll
        for (unsigned v = 0; v < KS; ++v) { t1[v].x *= fe; t1[v].y *= fo; }
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
      if (nt) s16_store_d_nt<KS>(op.parent + ubase, lane, t1); else s16_store_d<KS>(op.parent + ubase, lane, t1);
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
          double * unit = op.parent + ((size_t)blk * R + r) * UNIT;
          double2 t[KS];
          s16_load_d<KS>(unit, lane, t);
#pragma unroll
          for (unsigned v = 0; v < KS; ++v) { t[v].x *= fe; t[v].y *= fo; }
          s16_store_d<KS>(unit, lane, t);
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

// ---------------------------------------------------------------------------
// Whole traversals in one launch (PlanView, engine.h; see k_traverse_s20): chains of operations
// with the handed-over block in registers (the D layout of a result is the B layout of an
// operand here as well), the tables of a whole chain in LDS, a workgroup walking all chains of
// the schedule with the same site blocks.  Four rate categories (the register-resident block
// needs the rate count at compile time); other rate counts keep the level schedule.
// grid = gx <= CUs, block = 512, dynamic LDS = the largest chain area of the schedule.
// ---------------------------------------------------------------------------
constexpr unsigned S16_CHAIN_WAVES = 8;
constexpr unsigned S16_CHAIN_MAX = 8;
constexpr unsigned S16_CHAIN_LDS = 20480;             // doubles: the whole LDS of a CU

// one operation for one site block; X: the handed-over operand on entry (carried != 0), the result on
// exit; xe / xo: the scaler counts that go with X (per rate with RS, else element 0)
// WIDE: a child may be a "wide tip" (kernels_repeats.hpp: a node known per class of sites) -- neither vector nor byte
// codes; pfragN holds its 32-bit class codes, lutN its row table [rate][class][S], childN_index the rows, scalerN its
// counts per class
template <unsigned KS, unsigned RT, bool RS, bool WIDE = false>
__device__ inline void s16_chain_op(const OpDesc & op, unsigned carried, double2 X[RT][KS],
                                    const double * s1, const double * s2, unsigned S, unsigned lut_codes,
                                    bool lut_lds, unsigned blk, unsigned lane,
                                    unsigned (&xe)[RS ? RT : 1], unsigned (&xo)[RS ? RT : 1], bool nt, bool ntl = false,
                                    bool store = true,   // store: false = handed on in registers only (PlanOp::flags bit 0)
                                    unsigned wide_lds = 0)   // PlanOp::flags bit 1 / 2: the rows of wide tip 1 / 2 are staged at s1 / s2
{
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const unsigned q = lane >> 4, n = lane & 15;
  const size_t site0 = (size_t)blk * S20_BS + 2 * n;
  unsigned c1e = 0, c1o = 0, c2e = 0, c2o = 0;
  const unsigned * w1 = (WIDE && !op.clv1 && !op.codes1) ? reinterpret_cast<const unsigned *>(op.pfrag1) : nullptr;
  const unsigned * w2 = (WIDE && !op.clv2 && !op.codes2) ? reinterpret_cast<const unsigned *>(op.pfrag2) : nullptr;
  if (op.codes1) { c1e = op.codes1[site0]; c1o = op.codes1[site0 + 1]; }
  else if (w1) { c1e = w1[site0]; c1o = w1[site0 + 1]; }
  if (op.codes2) { c2e = op.codes2[site0]; c2o = op.codes2[site0 + 1]; }
  else if (w2) { c2e = w2[site0]; c2o = w2[site0 + 1]; }
  const bool scaling = op.parent_scaler != nullptr;
  const double * l1 = lut_lds ? s1 : op.lut1, * l2 = lut_lds ? s2 : op.lut2;
  // where a child's scaler counts of the lane's two sites are: per site, or (wide tip) per class
  const size_t k1e = w1 ? c1e : site0, k1o = w1 ? c1o : site0 + 1, k2e = w2 ? c2e : site0, k2o = w2 ? c2o : site0 + 1;
  int small_e = 1, small_o = 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    const size_t ubase = ((size_t)blk * RT + r) * UNIT;
    double2 t1[KS], t2[KS];
    if (carried == 1) s16_child_regs<KS>(X[r], s1 + r * s16_fr(KS), lane, t1);
    else if (w1) s16_child_tip<KS>(((wide_lds & 2u) ? s1 : op.lut1) + (size_t)r * op.child1_index * S, c1e, c1o, q, S, t1);
    else if (!op.codes1) s16_child_inner<KS>(op.clv1 + ubase, s1 + r * s16_fr(KS), lane, t1, ntl);
    else s16_child_tip<KS>(l1 + (size_t)r * lut_codes * S, c1e, c1o, q, S, t1);
    if (carried == 2) s16_child_regs<KS>(X[r], s2 + r * s16_fr(KS), lane, t2);
    else if (w2) s16_child_tip<KS>(((wide_lds & 4u) ? s2 : op.lut2) + (size_t)r * op.child2_index * S, c2e, c2o, q, S, t2);
    else if (!op.codes2) s16_child_inner<KS>(op.clv2 + ubase, s2 + r * s16_fr(KS), lane, t2, ntl);
    else s16_child_tip<KS>(l2 + (size_t)r * lut_codes * S, c2e, c2o, q, S, t2);
    int re = 1, ro = 1;
#pragma unroll
    for (unsigned v = 0; v < KS; ++v)
    {
      X[r][v].x = t1[v].x * t2[v].x;
      X[r][v].y = t1[v].y * t2[v].y;
      re &= (X[r][v].x < SCALE_THRESHOLD);       // rows >= S are zero: they never veto
      ro &= (X[r][v].y < SCALE_THRESHOLD);
    }
    if (!RS)
    {
      small_e &= re;
      small_o &= ro;
      continue;
    }
    unsigned ce = 0, co = 0;
    if (scaling)
    {
      const int se = s20_and_q(re), so = s20_and_q(ro);
      const double fe = se ? SCALE_FACTOR : 1.0, fo = so ? SCALE_FACTOR : 1.0;
#pragma unroll
      for (unsigned v = 0; v < KS; ++v) { X[r][v].x *= fe; X[r][v].y *= fo; }
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
    if (!store) { }
    else if (nt) s16_store_d_nt<KS>(op.parent + ubase, lane, X[r]); else s16_store_d<KS>(op.parent + ubase, lane, X[r]);
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
    for (unsigned v = 0; v < KS; ++v) { X[r][v].x *= fe; X[r][v].y *= fo; }
    if (!store) continue;
    if (nt) s16_store_d_nt<KS>(op.parent + ((size_t)blk * RT + r) * UNIT, lane, X[r]);
    else s16_store_d<KS>(op.parent + ((size_t)blk * RT + r) * UNIT, lane, X[r]);
  }
  unsigned ce = 0, co = 0;
  if (scaling && q == 0)
  {
    ce = small_e ? 1u : 0u;
    co = small_o ? 1u : 0u;
    if (op.scaler1)
    {
      if (carried == 1) { ce += xe[0]; co += xo[0]; }
      else { ce += op.scaler1[k1e]; co += op.scaler1[k1o]; }
    }
    if (op.scaler2)
    {
      if (carried == 2) { ce += xe[0]; co += xo[0]; }
      else { ce += op.scaler2[k2e]; co += op.scaler2[k2o]; }
    }
    op.parent_scaler[site0] = ce;
    op.parent_scaler[site0 + 1] = co;
  }
  xe[0] = ce;
  xo[0] = co;
}

template <unsigned KS, unsigned RT, bool RS, bool WIDE = false>
__global__ __launch_bounds__(64 * S16_CHAIN_WAVES, 1) void k_traverse_s16(PlanView plan, unsigned chain_begin,
                                                                           unsigned chain_end, unsigned S,
                                                                           unsigned Sp, unsigned nt_flags)
{
  extern __shared__ double lds[];
  const bool nt = (nt_flags & 2u) != 0, ntl = (nt_flags & 4u) != 0;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wstride = gridDim.x * S16_CHAIN_WAVES;
  const PlanOp * plan_ops_;
  const PlanChain * plan_chains_;
  plan_bases(plan, plan_ops_, plan_chains_);
  bool first_fill = true;
  for (unsigned c = chain_begin + blockIdx.y; c < chain_end; c += gridDim.y)
  {
    const PlanChain ch = plan_fetch(plan_chains_ + c);
    // site blocks and tip tables of the partition this chain belongs to
    const unsigned nblk = ch.extent, lut_codes = ch.lut_codes;
    const bool lut_lds = (ch.flags & 1u) != 0;
    if (!first_fill) __syncthreads();
    first_fill = false;
    for (unsigned i = 0; i < ch.len; ++i)
    {
      const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
      // (a wide tip has its rows in LDS when they are few -- PlanOp::flags bit 1 / 2 --, else they are gathered from memory)
      if (WIDE && !po.d.clv1 && !po.d.codes1) { if (po.flags & 2u) staged_copy<8>(lds + po.slot1, po.d.lut1, RT * po.d.child1_index * S); }
      else if (!po.d.codes1) s16_fill_frags<KS>(lds + po.slot1, po.d.pmat1, RT, S, Sp);
      else if (lut_lds)
        staged_copy<8>(lds + po.slot1, po.d.lut1, RT * lut_codes * S);
      if (WIDE && !po.d.clv2 && !po.d.codes2) { if (po.flags & 4u) staged_copy<8>(lds + po.slot2, po.d.lut2, RT * po.d.child2_index * S); }
      else if (!po.d.codes2) s16_fill_frags<KS>(lds + po.slot2, po.d.pmat2, RT, S, Sp);
      else if (lut_lds)
        staged_copy<8>(lds + po.slot2, po.d.lut2, RT * lut_codes * S);
    }
    __syncthreads();
    for (unsigned blk = blockIdx.x * S16_CHAIN_WAVES + wave; blk < nblk; blk += wstride)
    {
      double2 X[RT][KS];
      unsigned xe[RS ? RT : 1] = {}, xo[RS ? RT : 1] = {};
#pragma unroll 1
      for (unsigned i = 0; i < ch.len; ++i)
      {
        const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
        s16_chain_op<KS, RT, RS, WIDE>(po.d, i ? po.carried : 0u, X, lds + po.slot1, lds + po.slot2, S, lut_codes,
                                 lut_lds, blk, lane, xe, xo, nt, ntl, !(po.flags & 1u), WIDE ? po.flags : 0u);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// edge / root log-likelihood.  grid = nblocks, block = 256; dynamic LDS = R * KS * 64 doubles
// ---------------------------------------------------------------------------
template <unsigned KS>
__global__ __launch_bounds__(256) void k_edge_lnl_s16(ModelView mv, ParamIdx fidx, NodeRef parent, NodeRef child,
                                                      const double * pmat, const double * lut, unsigned lut_codes,
                                                      const unsigned * ps, const unsigned * cs,
                                                      const unsigned * weights, const int * invariant,
                                                      const unsigned long long * tipmap,
                                                      unsigned N, unsigned nblk, unsigned R,
                                                      double * persite, ReduceOut block_out, unsigned rate_scalers)
{
  extern __shared__ double frag[];
  __shared__ double scratch[4];
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const unsigned S = mv.S;
  if (pmat && !child.codes) s16_fill_frags<KS>(frag, pmat, R, S, mv.Sp);
  __syncthreads();

  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned wstride = gridDim.x * 4;
  double acc = 0.0;

  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += wstride)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    SiteSide sd = {0u, 0u, 0u, 0u};
    if (rate_scalers)
    {
      // every lane needs the per-rate factors of its two sites (padding sites carry zeros)
      sd.cnt_e = rate_min_count(ps, cs, site0, R);
      sd.cnt_o = rate_min_count(ps, cs, site0 + 1, R);
      if (site0 < N) sd.w_e = weights[site0];
      if (site0 + 1 < N) sd.w_o = weights[site0 + 1];
    }
    else sd = load_site_side(ps, cs, weights, site0, N, q == 0);
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
      const size_t ubase = ((size_t)blk * R + r) * UNIT;
      const unsigned fi = fidx.v[r];
      const double * pi = mv.freqs(fi);
      double2 t[KS], pv[KS];
      if (!pmat)
      {
#pragma unroll
        for (unsigned v = 0; v < KS; ++v) t[v] = make_double2(1.0, 1.0);
      }
      else if (child.codes) s16_child_tip<KS>(lut + (size_t)r * lut_codes * S, cce, cco, q, S, t);
      else s16_child_inner<KS>(child.clv + ubase, frag + r * s16_fr(KS), lane, t);
      if (parent.codes) s16_tip_d<KS>(pme, pmo, q, S, pv);
      else s16_load_d<KS>(parent.clv + ubase, lane, pv);
      double le = 0.0, lo = 0.0;
#pragma unroll
      for (unsigned v = 0; v < KS; ++v)
      {
        const unsigned i = 4 * v + q;
        const double f = (i < S) ? pi[i] : 0.0;
        le += f * pv[v].x * t[v].x;
        lo += f * pv[v].y * t[v].y;
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
        const double l = site_loglh(site_e, sd.cnt_e, inv_e);
        if (persite) persite[site0] = l;
        acc += l * (double)sd.w_e;
      }
      if (site0 + 1 < N)
      {
        const double l = site_loglh(site_o, sd.cnt_o, inv_o);
        if (persite) persite[site0 + 1] = l;
        acc += l * (double)sd.w_o;
      }
    }
  }
  const double tot = block_sum_256(acc, scratch);
  grid_reduce_finish1(tot, block_out, scratch);
}

// ---------------------------------------------------------------------------
// sumtable preparation: eigen-basis matrices in the [r][row][Sp] form the partials kernel
// consumes, plus their tip lookup tables [r][code][S]
//   Lm[r][k][i] = pi_i V[i][k],   Rm[r][k][j] = V^-1[k][j]
// grid = R, block = 256
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumtable_prep_s16(ModelView mv, ParamIdx params,
                                                           const unsigned long long * tipmap,
                                                           unsigned lut_codes, bool want_lut,
                                                           double * Lm, double * Rm, double * lutL, double * lutR)
{
  const unsigned S = mv.S, Sp = mv.Sp;
  const unsigned r = blockIdx.x, pi_ = params.v[r];
  const double * pi = mv.freqs(pi_), * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_);
  double * L = Lm + (size_t)r * S * Sp, * Rr = Rm + (size_t)r * S * Sp;
  for (unsigned e = threadIdx.x; e < S * Sp; e += blockDim.x)
  {
    const unsigned k = e / Sp, i = e % Sp;
    L[e] = (i < S) ? pi[i] * V[i * Sp + k] : 0.0;
    Rr[e] = (i < S) ? Vi[k * Sp + i] : 0.0;
  }
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

// --- launchers -------------------------------------------------------------

#define PLLHIP_DISPATCH_KS(ks, CALL) \
  do { switch (ks) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
                     case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); } } while (0)

static unsigned s16_ks(const Engine * e) { return (e->S + 3) / 4; }
// A-fragment doubles per (child, rate): MT x KS x 64
static unsigned s16_frags(const Engine * e) { const unsigned ks = s16_ks(e); return ((ks + 3) / 4) * ks * 64; }
// can the family carry this alphabet and rate count?  (both children's fragments of all rates in LDS)
static bool s16_supported(unsigned S, unsigned R)
{
  const unsigned ks = (S + 3) / 4;
  return S >= 2 && S <= 32 && (size_t)2 * R * ((ks + 3) / 4) * ks * 64 * sizeof(double) <= (size_t)S16_MAX_LDS_BYTES;
}

static unsigned s16_grid(const Engine * e, unsigned blocks_per_cu)
{
  const unsigned need = (e->nblk + 3) / 4;
  return std::max(1u, std::min(need, e->cu_count * blocks_per_cu));
}

// LDS doubles per child table: its A fragments, or its tip lookup table when that is small enough
static unsigned s16_table(const Engine * e)
{
  const unsigned frags = e->R * s16_frags(e), lut = e->R * e->lut_codes * e->S;
  return (e->coded_tips && lut <= S16_LUT_LDS) ? std::max(frags, lut) : frags;
}

// beyond 16 states with many rates the tables pass the 64 KiB a kernel gets without asking
static int s16_allow_lds(Engine * e)
{
  static bool attr_set_dev[64] = {false};
  bool & attr_set = attr_set_dev[e->device & 63];
  if (attr_set) return PLL_SUCCESS;
#define PLLHIP_ATTR(KK) \
  do { \
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partials_s16<KK>), hipFuncAttributeMaxDynamicSharedMemorySize, S16_MAX_LDS_BYTES)); \
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_lnl_s16<KK>), hipFuncAttributeMaxDynamicSharedMemorySize, S16_MAX_LDS_BYTES)); \
  } while (0)
  PLLHIP_ATTR(5); PLLHIP_ATTR(6); PLLHIP_ATTR(7); PLLHIP_ATTR(8);
#undef PLLHIP_ATTR
  attr_set = true;
  return PLL_SUCCESS;
}

static int launch_partials_s16(Engine * e, const OpBatch & batch, unsigned nops, bool traversal = true)
{
  static const int env_nt = getenv("PLLHIP_S16_NT") ? atoi(getenv("PLLHIP_S16_NT")) : 2;
  const unsigned nt = traversal && env_nt ? 1u : 0u;
  const unsigned table = s16_table(e);
  if (sizeof(double) * 2 * table > 64 * 1024 && !s16_allow_lds(e)) return PLL_FAILURE;
#define PLLHIP_CALL(KK) \
  hipLaunchKernelGGL(k_partials_s16<KK>, dim3(s16_grid(e, 8), nops), dim3(256), sizeof(double) * 2 * table, e->stream, \
                     batch, e->nblk, e->R, e->S, e->Sp, e->lut_codes, table, e->rate_scalers ? 1u : 0u, nt)
  PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// one-launch traversals: four rate categories
// (per-site scaling keeps the block of ALL rates in registers until the vote: 4 x KS x 4 VGPRs; beyond
// 20 states that spills -- 51 / 135 / 302 registers at 24 / 28 / 32 states -- and the level schedule with its
// store-unscaled-and-fix-up kernel is the faster form at 32 states (500 k sites, 50 taxa: 12.8 against 14.7 ms;
// 24 / 28 states: chains 6.4 / 10.1 against 8.0 / 12.8 ms).  Per-rate scalers settle a rate at a time: no limit.
// PLLHIP_S16_CHAIN_KS: largest KS = ceil(S / 4) that takes chains with per-site scaling.)
// (Round 4, with the tip tables of 28 and 32 states in LDS and the operands of the next rate requested ahead in the
// level-schedule kernel: 28 states 8.7 ms by levels against 9.7 in chains -- chains up to 24 states now, and up to 28 for a
// partition that asks for site repeats, which live in the chain schedule.)
static bool chains_supported_s16(const Engine * e)
{
  static const unsigned env_ks = getenv("PLLHIP_S16_CHAIN_KS") ? (unsigned)atoi(getenv("PLLHIP_S16_CHAIN_KS")) : 0u;
  const unsigned max_ks = env_ks ? env_ks : (e->site_repeats ? 7u : 6u);
  return e->R == 4 && (e->rate_scalers || (e->S + 3) / 4 <= max_ks);
}

static bool s16_chain_lut_lds(const Engine * e) { return e->R * e->lut_codes * e->S <= S16_LUT_LDS; }

// LDS doubles of one child's table in a chain (0: tip table gathered from global memory)
static unsigned s16_chain_slot(const Engine * e, bool tip)
{
  if (!tip) return e->R * s16_frags(e);
  return s16_chain_lut_lds(e) ? ((e->R * e->lut_codes * e->S + 7u) & ~7u) : 0u;
}

// wide: the schedule may hold wide tips (site repeats; per-site scaling only)
static int launch_traverse_s16(Engine * e, const PlanView & plan, unsigned lds_doubles, unsigned extent, unsigned chain_begin,
                               unsigned chain_end, unsigned rows, unsigned row_wgs_per_cu = 0, bool wide = false)
{
  const size_t lds = sizeof(double) * lds_doubles;
  const unsigned need = (extent + S16_CHAIN_WAVES - 1) / S16_CHAIN_WAVES;
  const unsigned gx = round_grid(e, std::max(1u, std::min(need, e->cu_count)), rows, row_wgs_per_cu ? row_wgs_per_cu : 4u,
                                 (extent + 16u * S16_CHAIN_WAVES - 1) / (16u * S16_CHAIN_WAVES));
  static const int env_nt = getenv("PLLHIP_S16_NT") ? atoi(getenv("PLLHIP_S16_NT")) : 2;   // stores and loads past the caches: 3 - 15 % faster
  const unsigned nt_flags = (env_nt ? 2u : 0u) | (env_nt == 2 ? 4u : 0u);
  static bool attr_set_dev[64] = {false};
  bool & attr_set = attr_set_dev[e->device & 63];
  const int cap = (int)(sizeof(double) * S16_CHAIN_LDS);
#define PLLHIP_ATTR(KK) \
  do { \
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_traverse_s16<KK, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_traverse_s16<KK, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
    PLLHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_traverse_s16<KK, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  } while (0)
  if (!attr_set)
  {
    PLLHIP_ATTR(1); PLLHIP_ATTR(2); PLLHIP_ATTR(3); PLLHIP_ATTR(4);
    PLLHIP_ATTR(5); PLLHIP_ATTR(6); PLLHIP_ATTR(7); PLLHIP_ATTR(8);
    attr_set = true;
  }
#undef PLLHIP_ATTR
#define PLLHIP_CALL(KK) \
  do { \
    if (e->rate_scalers) \
      hipLaunchKernelGGL((k_traverse_s16<KK, 4, true>), dim3(gx, std::max(1u, rows)), dim3(64 * S16_CHAIN_WAVES), lds, e->stream, \
                         plan, chain_begin, chain_end, e->S, e->Sp, nt_flags); \
    else if (wide) \
      hipLaunchKernelGGL((k_traverse_s16<KK, 4, false, true>), dim3(gx, std::max(1u, rows)), dim3(64 * S16_CHAIN_WAVES), lds, e->stream, \
                         plan, chain_begin, chain_end, e->S, e->Sp, nt_flags); \
    else \
      hipLaunchKernelGGL((k_traverse_s16<KK, 4, false>), dim3(gx, std::max(1u, rows)), dim3(64 * S16_CHAIN_WAVES), lds, e->stream, \
                         plan, chain_begin, chain_end, e->S, e->Sp, nt_flags); \
  } while (0)
  PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_edge_lnl_s16(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                               const NodeRef & parent, const NodeRef & child,
                               const double * pm, const double * lut,
                               const unsigned * ps, const unsigned * cs,
                               double * persite, unsigned nblocks)
{
  if (sizeof(double) * e->R * s16_frags(e) > 64 * 1024 && !s16_allow_lds(e)) return PLL_FAILURE;
#define PLLHIP_CALL(KK) \
  hipLaunchKernelGGL(k_edge_lnl_s16<KK>, dim3(nblocks), dim3(256), sizeof(double) * e->R * s16_fr(KK), e->stream, \
                     mv, fidx, parent, child, pm, lut, e->lut_codes, ps, cs, e->d_weights, e->d_invariant, \
                     e->d_tipmap, e->N, e->nblk, e->R, persite, reduce_out(e), e->rate_scalers ? 1u : 0u)
  PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_sumtable_s16(Engine * e, const ModelView & mv, const ParamIdx & params,
                               const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  // scratch: Lm | Rm | lutL | lutR
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
    hipLaunchKernelGGL(k_sumtable_prep_s16, dim3(e->R), dim3(256), 0, e->stream,
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
  return launch_partials_s16(e, batch, 1, false);
}

static int launch_derivatives_s16(Engine * e, const ModelView & mv, const ParamIdx & params,
                                  const TrialLengths & tl, unsigned count,
                                  const double * d_sum, const unsigned * ps, const unsigned * cs,
                                  unsigned nblocks)
{
#define PLLHIP_CALL(KK) \
  hipLaunchKernelGGL((k_derivatives_mfma<KK, 0>), dim3(nblocks), dim3(256), sizeof(double) * e->R * KK * 64, e->stream, \
                     mv, params, tl, count, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->nblk, e->R, \
                     reduce_out(e), e->rate_scalers ? 1u : 0u)
  PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

} // namespace pllhip
