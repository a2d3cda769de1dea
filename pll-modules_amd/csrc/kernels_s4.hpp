// kernels_s4.hpp -- 4-state (DNA) kernel family, fp64 VALU.
//
// Roofline: a 4-state site-update moves 24*4 + 12/R bytes and needs 68 flops
// (AI 0.69 flop/B): purely HBM-bound, so the design goal is nothing but
// perfectly streaming access:
//   * partials: two lanes per (site, rate) column, 16 bytes each, so every wave
//     load/store is one contiguous 1 KiB run; the halves meet through one xor-1
//     lane exchange per output (k_partials_s4);
//   * reductions (edge lnL, derivatives): one lane per column, four grid-stride
//     steps per trip so that 8 loads per lane are in flight;
//   * a lane's rate (and half) is invariant under the grid stride, so its
//     P-matrix entries stay in VGPRs for the whole kernel;
//   * the per-site scaling vote and the sum over rates are butterfly shuffles
//     over the lanes of a site (R is a power of two <= 16 in this family, so
//     all index arithmetic is shifts and masks);
//   * coded tips cost 1 byte per site: the child term is a 32-byte gather from
//     the 2 KiB per-matrix lookup table (L1/L2 resident).
#pragma once

#include "kernels_common.hpp"
#include "engine.h"

namespace pllhip {

struct d4 { double x, y, z, w; };

__device__ inline d4 load4(const double * p)
{
  const double4 v = *reinterpret_cast<const double4 *>(p);
  return d4{v.x, v.y, v.z, v.w};
}

__device__ inline void store4(double * p, const d4 & v)
{
  *reinterpret_cast<double4 *>(p) = make_double4(v.x, v.y, v.z, v.w);
}

__device__ inline double dot4(const double * row, const d4 & c)
{
  return row[0] * c.x + row[1] * c.y + row[2] * c.z + row[3] * c.w;
}

// lane ^ 1 / lane ^ 2 within a quad: DPP quad permutes, no LDS round trip
template <int CTRL>
__device__ inline double dpp_quad(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ inline double group_sum(double v, unsigned R)
{
  if (R >= 2) v += dpp_quad<0xB1>(v);            // quad_perm [1,0,3,2]
  if (R >= 4) v += dpp_quad<0x4E>(v);            // quad_perm [2,3,0,1]
  for (unsigned off = 4; off < R; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline d4 tip_value4(unsigned long long mask)
{
  return d4{(double)(mask & 1ULL), (double)((mask >> 1) & 1ULL),
            (double)((mask >> 2) & 1ULL), (double)((mask >> 3) & 1ULL)};
}

// Partials, lane-pair mapping: two lanes per (site, rate) column, each owning
// two of the four states.  Lane g loads and stores the 16-byte half-vector at
// byte g*16 of the CLV, so every wave instruction moves one contiguous 1 KiB run.
//   own_a = sum_b P[2h+a][2h+b] c[2h+b]        (this lane's outputs, own inputs)
//   oth_a = sum_b P[2(1-h)+a][2h+b] c[2h+b]    (partner's outputs, own inputs)
//   out_a = own_a + partner's oth_a            (one xor-1 lane exchange per value)
// The lane's half h and rate r are invariant under the grid stride (a multiple of
// 2R), so its 16 P-matrix entries stay in 32 VGPRs.
// grid = (gx, ops), block = 256.
struct HalfP { double own[4], oth[4]; };   // [a*2+b]

// rate stride of a lookup table [rate][16 codes][4] staged in LDS: 68 doubles -- with 64 all
// rates of a code share their banks
constexpr unsigned S4_LUT_RS = 68;

__device__ inline HalfP s4_load_half_p(const double * pmat, unsigned r, unsigned h)
{
  HalfP q;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
    {
      q.own[a * 2 + b] = pmat[r * 16 + (2 * h + a) * 4 + 2 * h + b];
      q.oth[a * 2 + b] = pmat[r * 16 + (2 * (1 - h) + a) * 4 + 2 * h + b];
    }
  return q;
}

// value of the partner lane (lane ^ 1): a DPP quad permute, no LDS round trip
__device__ inline double swap_pair(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ inline double2 s4_half_matvec(const HalfP & q, const double2 c)
{
  const double own0 = q.own[0] * c.x + q.own[1] * c.y, own1 = q.own[2] * c.x + q.own[3] * c.y;
  const double oth0 = q.oth[0] * c.x + q.oth[1] * c.y, oth1 = q.oth[2] * c.x + q.oth[3] * c.y;
  return make_double2(own0 + swap_pair(oth0), own1 + swap_pair(oth1));
}

// 1 if any of the `group` lanes (aligned, power of two <= 32) of this lane's site votes 1
__device__ inline int group_any(int vote, unsigned lane, unsigned group)
{
  const unsigned long long b = __ballot(vote);
  const unsigned long long m = (group >= 64) ? ~0ULL : ((1ULL << group) - 1ULL);
  return ((b >> (lane & ~(group - 1u))) & m) != 0ULL;
}

// vectors are written once and read (if at all) once: accesses that do not allocate in the caches
template <bool NTL>
__device__ inline double2 s4_ld2(const double * p)
{
  if (NTL)
  {
    typedef double nt_v2d __attribute__((ext_vector_type(2)));
    const nt_v2d w = __builtin_nontemporal_load(reinterpret_cast<const nt_v2d *>(p));
    return make_double2(w.x, w.y);
  }
  return *reinterpret_cast<const double2 *>(p);
}

__device__ inline void s4_st2_nt(double * p, const double2 & v)
{
  typedef double nt_v2d __attribute__((ext_vector_type(2)));
  nt_v2d w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<nt_v2d *>(p));
}

template <unsigned U>    // iterations issued per batch (loads first, arithmetic after)
__global__ __launch_bounds__(256) void k_partials_s4(OpBatch batch, unsigned N, unsigned R,
                                                     unsigned lut_codes)
{
  const OpDesc & op = batch.op[blockIdx.y];
  const unsigned lane = threadIdx.x & 63;
  const unsigned h = lane & 1u, r = (lane >> 1) & (R - 1);
  const unsigned group = 2 * R;                      // lanes per site
  const unsigned spi = 64 / group;                   // sites per wave iteration
  const unsigned rs = (unsigned)__ffs((int)R) - 1;   // R is a power of two in this family
  const unsigned long long total = 2ULL * N * R;     // half-columns

  // tip lookup tables (R x 16 codes x 4 doubles each) are staged in LDS: the
  // per-site gathers then cost LDS bank cycles instead of L1 address cycles.
  // Inner children stage their R 4x4 matrices the same way (one global load per
  // thread instead of sixteen scattered ones per lane: the prologue of a workgroup
  // that lives for a few chunks only).
  __shared__ double lut_s[2][16 * S4_LUT_RS];        // [rate][16 codes][4], rate stride padded (bank spread)
  __shared__ double pm_s[2][16 * 16];
  const bool lut_lds = lut_codes == 16;              // always true for DNA tip codes
  if (op.codes1) { if (lut_lds) for (unsigned e = threadIdx.x; e < R * 64; e += 256) lut_s[0][(e >> 6) * S4_LUT_RS + (e & 63)] = op.lut1[e]; }
  else for (unsigned e = threadIdx.x; e < R * 16; e += 256) pm_s[0][e] = op.pmat1[e];
  if (op.codes2) { if (lut_lds) for (unsigned e = threadIdx.x; e < R * 64; e += 256) lut_s[1][(e >> 6) * S4_LUT_RS + (e & 63)] = op.lut2[e]; }
  else for (unsigned e = threadIdx.x; e < R * 16; e += 256) pm_s[1][e] = op.pmat2[e];
  __syncthreads();
  HalfP p1 = {}, p2 = {};
  if (!op.codes1) p1 = s4_load_half_p(pm_s[0], r, h);
  if (!op.codes2) p2 = s4_load_half_p(pm_s[1], r, h);

  // A wave owns chunks of 64 consecutive sites = `group` iterations of 64
  // half-columns (1 KiB per child each).  The scaling vote of an iteration is
  // known inside the iteration (values are scaled before they are stored); the
  // 64 scaler counts of the chunk are written once, coalesced, at the end.
  const unsigned nchunks = (N + 63) / 64;
  const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  for (unsigned chunk = wave; chunk < nchunks; chunk += nwaves)
  {
    const unsigned long long hc0 = (unsigned long long)chunk * 64ULL * group + lane;
    unsigned scaled_mask = 0;                        // bit k: iteration k rescaled (same in a lane group)
    // the children's scaler counts of this lane's site: fetched now, needed after the chunk
    const unsigned long long nsc = (unsigned long long)chunk * 64ULL + lane;
    unsigned child_cnt = 0;
    if (op.parent_scaler && nsc < N)
      child_cnt = (op.scaler1 ? op.scaler1[nsc] : 0u) + (op.scaler2 ? op.scaler2[nsc] : 0u);
    // the tip codes of the chunk's 64 sites: one coalesced byte load per child (lane l holds
    // site l); an iteration picks its site's code with a lane shuffle
    const unsigned long long ncode = nsc < N ? nsc : 0ULL;
    const int cb1 = op.codes1 ? (int)op.codes1[ncode] : 0;
    const int cb2 = op.codes2 ? (int)op.codes2[ncode] : 0;
    for (unsigned k0 = 0; k0 < group; k0 += U)
    {
      double2 in1[U], in2[U];
      bool live[U];
#pragma unroll
      for (unsigned u = 0; u < U; ++u)
      {
        const unsigned long long gu = hc0 + (unsigned long long)(k0 + u) * 64ULL;
        const int src = (int)(((k0 + u) * spi + (lane >> (rs + 1))) & 63u);
        live[u] = (k0 + u < group) && gu < total;
        in1[u] = in2[u] = make_double2(0.0, 0.0);
        const int code1 = op.codes1 ? __shfl(cb1, src, 64) : 0;
        const int code2 = op.codes2 ? __shfl(cb2, src, 64) : 0;
        if (live[u])
        {
          if (!op.codes1) in1[u] = s4_ld2<true>(op.clv1 + gu * 2);
          else if (lut_lds)
            in1[u] = *reinterpret_cast<const double2 *>(&lut_s[0][r * S4_LUT_RS + code1 * 4 + 2 * h]);
          else
            in1[u] = *reinterpret_cast<const double2 *>(op.lut1 + ((size_t)r * lut_codes + code1) * 4 + 2 * h);
          if (!op.codes2) in2[u] = s4_ld2<true>(op.clv2 + gu * 2);
          else if (lut_lds)
            in2[u] = *reinterpret_cast<const double2 *>(&lut_s[1][r * S4_LUT_RS + code2 * 4 + 2 * h]);
          else
            in2[u] = *reinterpret_cast<const double2 *>(op.lut2 + ((size_t)r * lut_codes + code2) * 4 + 2 * h);
        }
      }
#pragma unroll
      for (unsigned u = 0; u < U; ++u)
      {
        if (k0 + u >= group) break;                  // wave-uniform
        const unsigned long long gu = hc0 + (unsigned long long)(k0 + u) * 64ULL;
        const double2 a = op.codes1 ? in1[u] : s4_half_matvec(p1, in1[u]);
        const double2 b = op.codes2 ? in2[u] : s4_half_matvec(p2, in2[u]);
        double2 v = make_double2(a.x * b.x, a.y * b.y);
        if (op.parent_scaler)
        {
          const int big = group_any(live[u] && !(v.x < SCALE_THRESHOLD && v.y < SCALE_THRESHOLD), lane, group);
          if (!big)
          {
            v.x *= SCALE_FACTOR;
            v.y *= SCALE_FACTOR;
            scaled_mask |= 1u << (k0 + u);
          }
        }
        if (live[u]) s4_st2_nt(op.parent + gu * 2, v);
      }
    }
    if (op.parent_scaler)
    {
      // lane l finishes site chunk*64 + l: it was handled in iteration l / spi by
      // lane group l % spi
      const unsigned src = (lane % spi) * group;
      const unsigned m = (unsigned)__shfl((int)scaled_mask, (int)src, 64);
      if (nsc < N) op.parent_scaler[nsc] = child_cnt + ((m >> (lane / spi)) & 1u);
    }
  }
}

// Operation chains (ChainBatch, engine.h), four rate categories: a wave keeps the 64-site
// chunk it has just computed in registers (8 half-columns per lane) and feeds it to the
// next operation of the chain, so the carried child is not read back from HBM.  The
// lookup tables / matrices of ALL operations of the chain are staged in LDS up front
// (5 KiB per operation); the waves of a workgroup never synchronise afterwards.  Values
// are scaled before they are handed on, i.e. exactly what k_partials_s4 stores and the
// next launch would have loaded.  grid = (gx, chains), block = 256,
// dynamic LDS = chain length x S4_CHAIN_OP_LDS doubles.
constexpr unsigned S4_CHAIN_MAX = 8;
// LDS doubles per operation: two tables [R][16][4] (rate stride padded to 68 doubles: with 64
// all rates of a code share their banks), two matrix sets [R][16]
constexpr unsigned s4_chain_op_lds(unsigned R) { return 2 * R * S4_LUT_RS + 2 * R * 16; }

// one operation of a chain for one 64-site chunk: X holds the handed-over operand on entry (when
// carried != 0) and the result on exit, xcnt the scaler count that goes with it.
// base: the operation's LDS tables
// WIDE: the operation may have wide tips (site repeats, tips kept per class); without them the instantiation carries
// none of their pointers and branches (one VGPR more and the traversal kernel loses its third wave per SIMD)
template <unsigned U, unsigned R, int NT = 0, bool WIDE = false>
__device__ inline void s4_chain_step(const OpDesc & op, unsigned carried, const double * base,
                                     double2 (&X)[2 * R], unsigned & xcnt,
                                     unsigned long long hc0, unsigned long long nsc, unsigned long long total,
                                     unsigned N, unsigned lane, unsigned r, unsigned h, bool store = true)
{
  // store: false = the result is handed to the next operation of the chain in registers only (PlanOp::flags bit 0)
  constexpr unsigned group = 2 * R, spi = 64 / group;
  constexpr unsigned T2 = R * S4_LUT_RS, M1 = 2 * R * S4_LUT_RS, M2 = 2 * R * S4_LUT_RS + R * 16;
  // a "wide tip" (kernels_repeats.hpp: a cherry known per class of sites): neither vector nor byte codes; pfrag
  // holds its 16-bit class codes, lut its table [rate][class][4], childN_index the classes
  const unsigned * w1 = (WIDE && !op.clv1 && !op.codes1) ? reinterpret_cast<const unsigned *>(op.pfrag1) : nullptr;
  const unsigned * w2 = (WIDE && !op.clv2 && !op.codes2) ? reinterpret_cast<const unsigned *>(op.pfrag2) : nullptr;
  const bool tip1 = op.codes1 || w1, tip2 = op.codes2 || w2;
  // (a wide tip of at most 16 classes -- every tip that is kept per class: 15 state masks -- has its rows in LDS, in
  // the place and layout of a coded tip's table: s4_chain_stage)
  const bool l1 = op.codes1 || (w1 && op.child1_index <= 16u), l2 = op.codes2 || (w2 && op.child2_index <= 16u);
  HalfP p1 = {}, p2 = {};
  if (!tip1) p1 = s4_load_half_p(base + M1, r, h);
  if (!tip2) p2 = s4_load_half_p(base + M2, r, h);
  unsigned child_cnt = 0;
  if (op.parent_scaler && nsc < N)
  {
    // (a wide tip's counts are kept per class, like its vector: indexed by the class code of the site)
    if (op.scaler1) child_cnt += (carried == 1) ? xcnt : w1 ? op.scaler1[w1[nsc]] : op.scaler1[nsc];
    if (op.scaler2) child_cnt += (carried == 2) ? xcnt : w2 ? op.scaler2[w2[nsc]] : op.scaler2[nsc];
  }
  unsigned scaled_mask = 0;
  // the tip codes of the chunk's 64 sites: one coalesced byte load per child (lane l holds
  // site l of the chunk); an iteration picks its site's code with a lane shuffle, so no
  // global load sits in front of the table lookup
  const unsigned long long ncode = nsc < N ? nsc : 0ULL;
  const int cb1 = op.codes1 ? (int)op.codes1[ncode] : w1 ? (int)w1[ncode] : 0;
  const int cb2 = op.codes2 ? (int)op.codes2[ncode] : w2 ? (int)w2[ncode] : 0;
#pragma unroll
  for (unsigned k0 = 0; k0 < group; k0 += U)
  {
    double2 in1[U], in2[U];
    bool live[U];
#pragma unroll
    for (unsigned u = 0; u < U; ++u)
    {
      const unsigned long long gu = hc0 + (unsigned long long)(k0 + u) * 64ULL;
      const int src = (int)((k0 + u) * spi + lane / group);       // this lane's site within the chunk
      live[u] = gu < total;
      in1[u] = in2[u] = make_double2(0.0, 0.0);
      const int code1 = tip1 ? __shfl(cb1, src, 64) : 0;
      const int code2 = tip2 ? __shfl(cb2, src, 64) : 0;
      if (live[u])
      {
        if (l1)
          in1[u] = *reinterpret_cast<const double2 *>(&base[r * S4_LUT_RS + code1 * 4 + 2 * h]);
        else if (w1)
          in1[u] = *reinterpret_cast<const double2 *>(&op.lut1[((size_t)r * op.child1_index + (unsigned)code1) * 4 + 2 * h]);
        else if (carried != 1) in1[u] = s4_ld2<NT == 2>(op.clv1 + gu * 2);
        if (l2)
          in2[u] = *reinterpret_cast<const double2 *>(&base[T2 + r * S4_LUT_RS + code2 * 4 + 2 * h]);
        else if (w2)
          in2[u] = *reinterpret_cast<const double2 *>(&op.lut2[((size_t)r * op.child2_index + (unsigned)code2) * 4 + 2 * h]);
        else if (carried != 2) in2[u] = s4_ld2<NT == 2>(op.clv2 + gu * 2);
      }
    }
#pragma unroll
    for (unsigned u = 0; u < U; ++u)
    {
      const unsigned k = k0 + u;
      const unsigned long long gu = hc0 + (unsigned long long)k * 64ULL;
      const double2 a = tip1 ? in1[u] : s4_half_matvec(p1, carried == 1 ? X[k] : in1[u]);
      const double2 b = tip2 ? in2[u] : s4_half_matvec(p2, carried == 2 ? X[k] : in2[u]);
      double2 v = make_double2(a.x * b.x, a.y * b.y);
      if (op.parent_scaler)
      {
        const int big = group_any(live[u] && !(v.x < SCALE_THRESHOLD && v.y < SCALE_THRESHOLD), lane, group);
        if (!big)
        {
          v.x *= SCALE_FACTOR;
          v.y *= SCALE_FACTOR;
          scaled_mask |= 1u << k;
        }
      }
      X[k] = v;
      if (live[u] && store)
      {
        if (NT != 0)
        {
          typedef double nt_v2d __attribute__((ext_vector_type(2)));
          nt_v2d w; w.x = v.x; w.y = v.y;
          __builtin_nontemporal_store(w, reinterpret_cast<nt_v2d *>(op.parent + gu * 2));
        }
        else *reinterpret_cast<double2 *>(op.parent + gu * 2) = v;
      }
    }
  }
  if (op.parent_scaler)
  {
    const unsigned src = (lane % spi) * group;
    const unsigned m = (unsigned)__shfl((int)scaled_mask, (int)src, 64);
    xcnt = child_cnt + ((m >> (lane / spi)) & 1u);
    if (nsc < N) op.parent_scaler[nsc] = xcnt;
  }
  else xcnt = 0;
}

// stage an operation's tables: [R][16][4] lookup tables (rate stride padded) / [R][16] matrices
template <unsigned R, bool WIDE = false>
__device__ inline void s4_chain_stage(const OpDesc & op, double * base)
{
  constexpr unsigned T2 = R * S4_LUT_RS, M1 = 2 * R * S4_LUT_RS, M2 = 2 * R * S4_LUT_RS + R * 16;
  const unsigned trow = (threadIdx.x >> 6) * S4_LUT_RS + (threadIdx.x & 63);
  // a wide tip (neither vector nor codes) of at most 16 classes: its row table [rate][classes][4] into the layout of a
  // coded tip's table
  const unsigned n1 = (WIDE && !op.clv1 && !op.codes1 && op.child1_index <= 16u) ? op.child1_index * 4u : 0u;
  const unsigned n2 = (WIDE && !op.clv2 && !op.codes2 && op.child2_index <= 16u) ? op.child2_index * 4u : 0u;
  if (op.codes1) { if (threadIdx.x < R * 64) base[trow] = op.lut1[threadIdx.x]; }
  else if (n1) { if (threadIdx.x < R * n1) base[(threadIdx.x / n1) * S4_LUT_RS + threadIdx.x % n1] = op.lut1[threadIdx.x]; }
  else if (threadIdx.x < R * 16) base[M1 + threadIdx.x] = op.pmat1[threadIdx.x];
  if (op.codes2) { if (threadIdx.x < R * 64) base[T2 + trow] = op.lut2[threadIdx.x]; }
  else if (n2) { if (threadIdx.x < R * n2) base[T2 + (threadIdx.x / n2) * S4_LUT_RS + threadIdx.x % n2] = op.lut2[threadIdx.x]; }
  else if (threadIdx.x < R * 16) base[M2 + threadIdx.x] = op.pmat2[threadIdx.x];
}

// (152 VGPRs, three waves per SIMD; forcing four costs 24 spilled registers and 25 % of the rate)
template <unsigned U, unsigned R>      // R in {1, 2, 4}; U <= 2R loads issued per batch
__global__ __launch_bounds__(256) void k_chain_s4(ChainBatch batch, unsigned N)
{
  constexpr unsigned group = 2 * R;
  constexpr unsigned S4_CHAIN_OP_LDS = s4_chain_op_lds(R);
  extern __shared__ double lds[];
  const unsigned first = batch.first[blockIdx.y], len = batch.len[blockIdx.y];
  const unsigned lane = threadIdx.x & 63;
  const unsigned h = lane & 1u, r = (lane >> 1) & (R - 1);
  const unsigned long long total = 2ULL * N * R;

  for (unsigned i = 0; i < len; ++i) s4_chain_stage<R>(batch.op[first + i], lds + i * S4_CHAIN_OP_LDS);
  __syncthreads();

  const unsigned nchunks = (N + 63) / 64;
  // a workgroup owns a contiguous range of chunks, its four waves interleave inside it
  // (measured equal to a grid-stride interleaving within noise: 3.81-3.87 ms on C2)
  const unsigned cbeg = (unsigned)(((unsigned long long)nchunks * blockIdx.x) / gridDim.x);
  const unsigned cend = (unsigned)(((unsigned long long)nchunks * (blockIdx.x + 1)) / gridDim.x);
  for (unsigned chunk = cbeg + (threadIdx.x >> 6); chunk < cend; chunk += 4)
  {
    const unsigned long long hc0 = (unsigned long long)chunk * 64ULL * group + lane;
    const unsigned long long nsc = (unsigned long long)chunk * 64ULL + lane;   // this lane's site for scalers
    double2 X[group];
    unsigned xcnt = 0;                                 // scaler count that goes with X
#pragma unroll 1
    for (unsigned i = 0; i < len; ++i)
      s4_chain_step<U, R, 2>(batch.op[first + i], i ? batch.carried[first + i] : 0u, lds + i * S4_CHAIN_OP_LDS,
                          X, xcnt, hc0, nsc, total, N, lane, r, h);
  }
}

// A whole traversal in one launch (PlanView, engine.h; see k_traverse_s20): a workgroup walks its
// range of chunks through every chain of the schedule, re-staging the tables between two chains.
// More workgroups than fit the chip: each finishes its range and makes room for the next, so the
// ranges in flight at any time are a slab of the alignment that moves through the whole tree.
// TRANS: an evaluate-only traversal (PlanOp::flags bit 0 = the result is handed on in registers only).  A template
// flag: as a run-time test in front of the stores it cost one VGPR, 169 instead of 168 -- and with that the third
// wave per SIMD (C2: 5.2 instead of 3.4 ms)
// (three waves per SIMD: 168 VGPRs)
template <unsigned U, unsigned R, int NT = 0, bool TRANS = false, bool WIDE = false>
__global__ __launch_bounds__(256, 3) void k_traverse_s4(PlanView plan, unsigned chain_begin, unsigned chain_end)
{
  constexpr unsigned group = 2 * R;
  constexpr unsigned S4_CHAIN_OP_LDS = s4_chain_op_lds(R);
  extern __shared__ double lds[];
  const unsigned lane = threadIdx.x & 63;
  const unsigned h = lane & 1u, r = (lane >> 1) & (R - 1);
  const PlanOp * plan_ops_;
  const PlanChain * plan_chains_;
  plan_bases(plan, plan_ops_, plan_chains_);
  bool first_fill = true;
  for (unsigned c = chain_begin + blockIdx.y; c < chain_end; c += gridDim.y)
  {
    const PlanChain ch = plan_fetch(plan_chains_ + c);
    // the sites of the partition this chain belongs to (a batched schedule holds several partitions)
    const unsigned N = ch.extent;
    const unsigned long long total = 2ULL * N * R;
    const unsigned nchunks = (N + 63) / 64;
    const unsigned cbeg = (unsigned)(((unsigned long long)nchunks * blockIdx.x) / gridDim.x);
    const unsigned cend = (unsigned)(((unsigned long long)nchunks * (blockIdx.x + 1)) / gridDim.x);
    if (!first_fill) __syncthreads();
    first_fill = false;
    for (unsigned i = 0; i < ch.len; ++i)
    {
      const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
      s4_chain_stage<R, WIDE>(po.d, lds + i * S4_CHAIN_OP_LDS);
    }
    __syncthreads();
    for (unsigned chunk = cbeg + (threadIdx.x >> 6); chunk < cend; chunk += 4)
    {
      const unsigned long long hc0 = (unsigned long long)chunk * 64ULL * group + lane;
      const unsigned long long nsc = (unsigned long long)chunk * 64ULL + lane;
      double2 X[group];
      unsigned xcnt = 0;
#pragma unroll 1
      for (unsigned i = 0; i < ch.len; ++i)
      {
        const PlanOp po = plan_fetch_op(plan_ops_ + ch.first + i);
        s4_chain_step<U, R, NT, WIDE>(po.d, i ? po.carried : 0u, lds + i * S4_CHAIN_OP_LDS,
                            X, xcnt, hc0, nsc, total, N, lane, r, h, TRANS ? !(po.flags & 1u) : true);
      }
    }
  }
}

// edge log-likelihood: lane = (site, rate).  grid = nblocks (<= REDUCE_BLOCKS)
__global__ __launch_bounds__(256) void k_edge_lnl_s4(ModelView mv, ParamIdx fidx,
                                                     NodeRef parent, NodeRef child,
                                                     const double * pmat, const double * lut,
                                                     unsigned lut_codes,
                                                     const unsigned * ps, const unsigned * cs,
                                                     const unsigned * weights, const int * invariant,
                                                     const unsigned long long * tipmap,
                                                     unsigned N, unsigned R,
                                                     double * persite, ReduceOut block_out)
{
  __shared__ double scratch[4];
  const unsigned long long total = (unsigned long long)N * R;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ULL;
  unsigned long long g = (unsigned long long)blockIdx.x * 256ULL + threadIdx.x;
  const unsigned r = (unsigned)(g & (R - 1));
  const unsigned rs = (unsigned)__ffs((int)R) - 1;
  const unsigned fi = fidx.v[r];
  const double * pi = mv.freqs(fi);
  const d4 f = {pi[0], pi[1], pi[2], pi[3]};
  const double pinv = mv.pinv()[fi];
  const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
  const double winv = (pinv > 0.0) ? mv.weights()[r] * pinv : 0.0;
  double P[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) P[q] = pmat[r * 16 + q];

  double acc = 0.0;
  const unsigned long long limit = (total + 63ULL) & ~63ULL;
  // four grid-stride iterations per trip: all loads of a trip are issued before
  // the first use, so each lane keeps 8 x 16-32 B in flight
  for (; g < limit; g += 4 * stride)
  {
    d4 cv[4], pv[4];
    bool live[4];
    unsigned long long nn[4];
    unsigned cnt[4], wgt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      const unsigned long long gu = g + u * stride;
      live[u] = gu < total;
      nn[u] = live[u] ? gu >> rs : 0;
      cv[u] = d4{0, 0, 0, 0};
      pv[u] = d4{0, 0, 0, 0};
      cnt[u] = wgt[u] = 0;
      if (live[u])
      {
        cv[u] = child.codes ? load4(lut + ((size_t)r * lut_codes + child.codes[nn[u]]) * 4)
                            : load4(child.clv + gu * 4);
        pv[u] = parent.codes ? tip_value4(tipmap[parent.codes[nn[u]]]) : load4(parent.clv + gu * 4);
        // needed only after the reduction over the rates: fetch them with the vectors, not after
        cnt[u] = (ps ? ps[nn[u]] : 0u) + (cs ? cs[nn[u]] : 0u);
        wgt[u] = weights[nn[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      if (g + u * stride >= limit) break;          // wave-uniform
      double lr = 0.0, inv = 0.0;
      if (live[u])
      {
        const d4 c = cv[u];
        const d4 a = child.codes ? c
                                 : d4{dot4(P, c), dot4(P + 4, c), dot4(P + 8, c), dot4(P + 12, c)};
        lr = wr * (f.x * pv[u].x * a.x + f.y * pv[u].y * a.y + f.z * pv[u].z * a.z + f.w * pv[u].w * a.w);
        if (winv > 0.0 && invariant && invariant[nn[u]] >= 0) inv = winv * pi[invariant[nn[u]]];
      }
      lr = group_sum(lr, R);
      inv = group_sum(inv, R);
      if (live[u] && r == 0)
      {
        const unsigned long long n = nn[u];
        const double l = site_loglh(lr, cnt[u], inv);
        if (persite) persite[n] = l;
        acc += l * (double)wgt[u];
      }
    }
  }
  const double tot = block_sum_256(acc, scratch);
  grid_reduce_finish1(tot, block_out, scratch);
}

// sumtable: lane = (site, rate)
__global__ __launch_bounds__(256) void k_sumtable_s4(ModelView mv, ParamIdx params,
                                                     NodeRef parent, NodeRef child,
                                                     const unsigned long long * tipmap,
                                                     unsigned N, unsigned R, double * sumtable)
{
  const unsigned long long total = (unsigned long long)N * R;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ULL;
  unsigned long long g = (unsigned long long)blockIdx.x * 256ULL + threadIdx.x;
  const unsigned r = (unsigned)(g & (R - 1));
  const unsigned rs = (unsigned)__ffs((int)R) - 1;
  const unsigned pi_ = params.v[r];
  const double * pi = mv.freqs(pi_), * V = mv.evecs(pi_), * Vi = mv.ievecs(pi_);
  double L[16], Rm[16];        // L[k][i] = pi_i V[i][k];  Rm[k][j] = Vinv[k][j]
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      L[k * 4 + i] = pi[i] * V[i * 4 + k];
      Rm[k * 4 + i] = Vi[k * 4 + i];
    }
  for (; g < total; g += stride)
  {
    const unsigned long long n = g >> rs;
    const d4 pv = parent.codes ? tip_value4(tipmap[parent.codes[n]]) : load4(parent.clv + g * 4);
    const d4 cv = child.codes ? tip_value4(tipmap[child.codes[n]]) : load4(child.clv + g * 4);
    const d4 out = {dot4(L, pv) * dot4(Rm, cv), dot4(L + 4, pv) * dot4(Rm + 4, cv),
                    dot4(L + 8, pv) * dot4(Rm + 8, cv), dot4(L + 12, pv) * dot4(Rm + 12, cv)};
    store4(sumtable + g * 4, out);
  }
}

// derivatives of -lnL at K trial branch lengths from ONE pass over the sumtable:
// lane = (site, rate); totals: df[0], ddf[0], df[1], ddf[1], ...
template <int K>
__global__ __launch_bounds__(256) void k_derivatives_s4(ModelView mv, ParamIdx params, TrialLengths tl,
                                                        const double * sumtable,
                                                        const unsigned * ps, const unsigned * cs,
                                                        const unsigned * weights, const int * invariant,
                                                        unsigned N, unsigned R, ReduceOut block_out)
{
  __shared__ double scratch[4];
  const unsigned long long total = (unsigned long long)N * R;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ULL;
  unsigned long long g = (unsigned long long)blockIdx.x * 256ULL + threadIdx.x;
  const unsigned r = (unsigned)(g & (R - 1));
  const unsigned rs = (unsigned)__ffs((int)R) - 1;
  const unsigned pi_ = params.v[r];
  const double pinv = mv.pinv()[pi_];
  const double rho = mv.rates()[r] / (1.0 - pinv);
  const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
  const double winv = (pinv > 0.0) ? mv.weights()[r] * pinv : 0.0;
  double e0[K][4], e1[K][4], e2[K][4];
#pragma unroll
  for (int j = 0; j < K; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const double lam = mv.evals(pi_)[k] * rho, ex = exp(lam * tl.t[j]);
      e0[j][k] = wr * ex;
      e1[j][k] = wr * ex * lam;
      e2[j][k] = wr * ex * lam * lam;
    }
  double df[K], ddf[K];
#pragma unroll
  for (int j = 0; j < K; ++j) df[j] = ddf[j] = 0.0;
  const unsigned long long limit = (total + 63ULL) & ~63ULL;
  // four grid-stride iterations per trip (loads first, then the arithmetic)
  for (; g < limit; g += 4 * stride)
  {
    d4 sv[4];
    bool live[4];
    unsigned wgt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      const unsigned long long gu = g + u * stride;
      live[u] = gu < total;
      sv[u] = live[u] ? load4(sumtable + gu * 4) : d4{0, 0, 0, 0};
      wgt[u] = live[u] ? weights[gu >> rs] : 0u;     // with the table, not behind the reduction
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      const unsigned long long gu = g + u * stride;
      if (gu >= limit) break;                        // wave-uniform
      const unsigned long long n = live[u] ? gu >> rs : 0;
      double inv = 0.0;
      if (live[u] && winv > 0.0 && invariant && invariant[n] >= 0)
        inv = winv * mv.freqs(pi_)[invariant[n]];
      inv = group_sum(inv, R);
      if (live[u] && r == 0 && inv > 0.0)
      {
        const unsigned cnt = (ps ? ps[n] : 0u) + (cs ? cs[n] : 0u);
        inv = (cnt <= 3) ? ldexp(inv, 256 * (int)cnt) : INFINITY;
      }
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        double A = dot4(e0[j], sv[u]), B = dot4(e1[j], sv[u]), C = dot4(e2[j], sv[u]);
        A = group_sum(A, R);
        B = group_sum(B, R);
        C = group_sum(C, R);
        if (live[u] && r == 0)
        {
          if (inv > 0.0) A += inv;
          const double w = (double)wgt[u], ba = B / A, ca = C / A;
          df[j] -= w * ba;
          ddf[j] += w * (ba * ba - ca);
        }
      }
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

// --- launchers -------------------------------------------------------------

static unsigned s4_grid(const Engine * e, unsigned cap_blocks)
{
  const unsigned long long total = (unsigned long long)e->N * e->R;
  const unsigned long long need = (total + 255ULL) / 256ULL;
  return (unsigned)std::max<unsigned long long>(1ULL, std::min<unsigned long long>(need, cap_blocks));
}

static int launch_partials_s4(Engine * e, const OpBatch & batch, unsigned nops)
{
  // one wave per 64-site chunk (grid-stride over chunks), 4 waves per block
  const unsigned nchunks = (e->N + 63) / 64;
  const unsigned gx = std::max(1u, std::min((nchunks + 3) / 4, e->cu_count * 16u));
  if (e->R >= 2)
    hipLaunchKernelGGL(k_partials_s4<4>, dim3(gx, nops), dim3(256), 0, e->stream,
                       batch, e->N, e->R, e->lut_codes);
  else
    hipLaunchKernelGGL(k_partials_s4<2>, dim3(gx, nops), dim3(256), 0, e->stream,
                       batch, e->N, e->R, e->lut_codes);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// (coded tips: through the 16-row tables of the family; tips that are vectors are read like inner children)
static bool chains_supported_s4(const Engine * e)
{
  return (e->R == 4 || e->R == 2 || e->R == 1) && (!e->coded_tips || e->lut_codes == 16);
}

static int launch_chains_s4(Engine * e, const ChainBatch & batch, unsigned nchains, unsigned longest)
{
  const unsigned nchunks = (e->N + 63) / 64;
  static const int env_bpc = getenv("PLLHIP_S4_CHAIN_BPC") ? atoi(getenv("PLLHIP_S4_CHAIN_BPC")) : 8;
  const unsigned gx = std::max(1u, std::min((nchunks + 3) / 4, e->cu_count * (unsigned)std::max(1, env_bpc)));
  const size_t lds = sizeof(double) * longest * s4_chain_op_lds(e->R);
  const dim3 grid(gx, nchains);
  if (e->R == 4)
    hipLaunchKernelGGL((k_chain_s4<4, 4>), grid, dim3(256), lds, e->stream, batch, e->N);
  else if (e->R == 2)
    hipLaunchKernelGGL((k_chain_s4<4, 2>), grid, dim3(256), lds, e->stream, batch, e->N);
  else
    hipLaunchKernelGGL((k_chain_s4<2, 1>), grid, dim3(256), lds, e->stream, batch, e->N);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// `extent`: sites of the largest partition the chains [chain_begin, chain_end) belong to
static int launch_traverse_s4(Engine * e, const PlanView & plan, unsigned longest, unsigned extent, unsigned chain_begin,
                              unsigned chain_end, unsigned rows, unsigned row_wgs_per_cu = 0, bool transient = false,
                              bool wide = false)
{
  const unsigned nchunks = (extent + 63) / 64;
  const size_t lds = sizeof(double) * longest * s4_chain_op_lds(e->R);
  // vectors are written once and read (if at all) once, by a later chain: stores and loads that do not
  // allocate in the caches are 5 % (1 M sites) to 17 % (100 k - 250 k sites) faster.  PLLHIP_S4_NT=0: plain.
  static const int env_nt = getenv("PLLHIP_S4_NT") ? atoi(getenv("PLLHIP_S4_NT")) : 1;
  const void * fn = nullptr;
#define PLLHIP_PICK_W(NT_, T_, W_) fn = e->R == 4 ? reinterpret_cast<const void *>(k_traverse_s4<4, 4, NT_, T_, W_>)   \
                                    : e->R == 2 ? reinterpret_cast<const void *>(k_traverse_s4<4, 2, NT_, T_, W_>) \
                                                : reinterpret_cast<const void *>(k_traverse_s4<2, 1, NT_, T_, W_>)
#define PLLHIP_PICK(NT_, T_) do { if (wide) PLLHIP_PICK_W(NT_, T_, true); else PLLHIP_PICK_W(NT_, T_, false); } while (0)
  if (env_nt) { if (transient) PLLHIP_PICK(2, true); else PLLHIP_PICK(2, false); }
  else { if (transient) PLLHIP_PICK(0, true); else PLLHIP_PICK(0, false); }
#undef PLLHIP_PICK
#undef PLLHIP_PICK_W
  // exactly the workgroups that are resident at once -- of the instantiation that is launched -- (measured on C2,
  // workgroups per CU: 2: 4.06 ms, 3 = resident: 3.58, 4: 4.12, 6: 3.68, 8: 3.82; one launch per round of chains: 3.85)
  static const int env_bpc = getenv("PLLHIP_S4_TRAVERSE_BPC") ? atoi(getenv("PLLHIP_S4_TRAVERSE_BPC")) : 0;
  int per_cu = env_bpc;
  if (per_cu <= 0)
  {
    PLLHIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
    per_cu = std::max(1, per_cu);
  }
  // a round of chains (rows > 1): the chains share the chip, eight workgroups per CU and chain as in k_chain_s4
  const unsigned gx = round_grid(e, std::max(1u, std::min((nchunks + 3) / 4, e->cu_count * (rows > 1 ? 8u : (unsigned)per_cu))), rows, row_wgs_per_cu ? row_wgs_per_cu : 8u);
  const dim3 grid(gx, std::max(1u, rows));
  void * args[] = {(void *)&plan, (void *)&chain_begin, (void *)&chain_end};
  PLLHIP_TRY(hipLaunchKernel(fn, grid, dim3(256), args, lds, e->stream));
  return PLL_SUCCESS;
}

static int launch_edge_lnl_s4(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                              const NodeRef & parent, const NodeRef & child,
                              const double * pm, const double * lut,
                              const unsigned * ps, const unsigned * cs,
                              double * persite, unsigned nblocks)
{
  hipLaunchKernelGGL(k_edge_lnl_s4, dim3(nblocks), dim3(256), 0, e->stream,
                     mv, fidx, parent, child, pm, lut, e->lut_codes, ps, cs,
                     e->d_weights, e->d_invariant, e->d_tipmap, e->N, e->R, persite, reduce_out(e));
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_sumtable_s4(Engine * e, const ModelView & mv, const ParamIdx & params,
                              const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  const unsigned gx = s4_grid(e, e->cu_count * 16u);
  hipLaunchKernelGGL(k_sumtable_s4, dim3(gx), dim3(256), 0, e->stream,
                     mv, params, parent, child, e->d_tipmap, e->N, e->R, d_sum);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

static int launch_derivatives_s4(Engine * e, const ModelView & mv, const ParamIdx & params,
                                 const TrialLengths & tl, unsigned count,
                                 const double * d_sum, const unsigned * ps, const unsigned * cs,
                                 unsigned nblocks)
{
#define PLLHIP_CALL(KK) \
  hipLaunchKernelGGL(k_derivatives_s4<KK>, dim3(nblocks), dim3(256), 0, e->stream, \
                     mv, params, tl, d_sum, ps, cs, e->d_weights, e->d_invariant, e->N, e->R, reduce_out(e))
  PLLHIP_DISPATCH_K(count, PLLHIP_CALL);
#undef PLLHIP_CALL
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

} // namespace pllhip
