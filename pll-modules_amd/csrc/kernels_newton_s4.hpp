// kernels_newton_s4.hpp -- Newton-Raphson on one branch where the data is, 4-state family.
//
// The loop of k_newton_mfma (kernels_s20.hpp: one launch per branch, all workgroups co-resident, in-launch reduction,
// the step rule of src/optimize/opt_algorithms.c:133-261 on the block that draws the last ticket, bounded waits) around
// the scan of k_derivatives_s4<1> (kernels_s4.hpp): lane = (site, rate), four grid-stride items per trip.  Same grid,
// same items per thread in the same order, same block and grid sums: the totals -- and with them the iterates -- are
// those of one blocking pll_compute_likelihood_derivatives call per iterate, bit for bit.
//
// NT > 0: a thread has at most NT trips and keeps what it reads -- sumtable entries, pattern weights, the invariant-site
// term of its sites -- in registers between the scans (16 B x 4 x NT of the table per thread: the whole table of a
// 500 k-site partition on 1 024 workgroups with NT = 2); later scans touch no memory but the model's few numbers.
// NT = 0: the table is streamed on every scan (kept for measurements: in one launch it is no faster than one blocking
// call per iterate, so the engine leaves partitions beyond NT = 2 to the host loop).
#pragma once

#include "kernels_common.hpp"
#include "kernels_s4.hpp"
#include "kernels_s20.hpp"
#include "kernels_s61.hpp"

namespace pllhip {

// gv: the workgroups the loop runs on (the launch, or this partition's run of a launch shared with other partitions)
template <unsigned NT>
__device__ inline void newton_loop_s4(const ModelView & mv, const ParamIdx & params, const NewtonParams & np,
                                      const double * sumtable,
                                      const unsigned * ps, const unsigned * cs,
                                      const unsigned * weights, const int * invariant,
                                      unsigned N, unsigned R, const ReduceOut & ro,
                                      NewtonControl * ctl, double * host_out,
                                      unsigned long long * host_flag, unsigned long long host_seq, GridView gv)
{
  __shared__ double scratch[4];
  __shared__ double s_x;
  __shared__ unsigned s_status;
  __shared__ double s_tot1;
  const unsigned long long total = (unsigned long long)N * R;
  const unsigned long long stride = (unsigned long long)gv.G * 256ULL;
  const unsigned long long g0 = (unsigned long long)gv.b * 256ULL + threadIdx.x;
  const unsigned r = (unsigned)(g0 & (R - 1));
  const unsigned rs = (unsigned)__ffs((int)R) - 1;
  const unsigned pi_ = params.v[r];
  const double pinv = mv.pinv()[pi_];
  const double rho = mv.rates()[r] / (1.0 - pinv);
  const double wr = mv.weights()[r] * ((pinv > 0.0) ? (1.0 - pinv) : 1.0);
  const double winv = (pinv > 0.0) ? mv.weights()[r] * pinv : 0.0;
  const unsigned long long limit = (total + 63ULL) & ~63ULL;
  double ev[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ev[k] = mv.evals(pi_)[k];
  constexpr unsigned NC = NT ? NT : 1;
  d4 c_sv[NC][4];
  double c_inv[NC][4];
  unsigned c_wgt[NC][4];

  double x = np.x0;
  if (gv.b == np.stall_block) return;                   // (fault injection: a workgroup that never arrives)
  for (unsigned it = 0; ; ++it)
  {
    double e0[4], e1[4], e2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const double lam = ev[k] * rho, ex = exp(lam * (np.xscale * x));
      e0[k] = wr * ex;
      e1[k] = wr * ex * lam;
      e2[k] = wr * ex * lam * lam;
    }
    double df = 0.0, ddf = 0.0;
    unsigned trip = 0;
    for (unsigned long long g = g0; g < limit; g += 4 * stride, ++trip)
    {
      d4 sv[4];
      double inv4[4];
      unsigned wgt[4];
      // (a trip beyond the NT the registers hold -- a grid smaller than the one the host sized NT for -- is streamed)
      const bool fetch = NT == 0 || it == 0 || trip >= NT;
      if (fetch)
      {
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
          const unsigned long long gu = g + u * stride;
          live[u] = gu < total;
          sv[u] = live[u] ? load4(sumtable + gu * 4) : d4{0, 0, 0, 0};
          wgt[u] = live[u] ? weights[gu >> rs] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
          const unsigned long long gu = g + u * stride;
          inv4[u] = 0.0;
          if (gu >= limit) continue;                       // wave-uniform
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
          inv4[u] = inv;
        }
        if (NT)
        {
#pragma unroll
          for (unsigned t = 0; t < NC; ++t)
            if (t == trip)
            {
#pragma unroll
              for (int u = 0; u < 4; ++u) { c_sv[t][u] = sv[u]; c_inv[t][u] = inv4[u]; c_wgt[t][u] = wgt[u]; }
            }
        }
      }
      else
      {
#pragma unroll
        for (unsigned t = 0; t < NC; ++t)
          if (t == trip)
          {
#pragma unroll
            for (int u = 0; u < 4; ++u) { sv[u] = c_sv[t][u]; inv4[u] = c_inv[t][u]; wgt[u] = c_wgt[t][u]; }
          }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const unsigned long long gu = g + u * stride;
        if (gu >= limit) break;                            // wave-uniform
        const bool live = gu < total;
        double A = dot4(e0, sv[u]), B = dot4(e1, sv[u]), C = dot4(e2, sv[u]);
        A = group_sum(A, R);
        B = group_sum(B, R);
        C = group_sum(C, R);
        if (live && r == 0)
        {
          if (inv4[u] > 0.0) A += inv4[u];
          const double w = (double)wgt[u], ba = B / A, ca = C / A;
          df -= w * ba;
          ddf += w * (ba * ba - ca);
        }
      }
    }
    const double t0 = block_sum_256(df, scratch);
    const double t1 = block_sum_256(ddf, scratch);
    if (threadIdx.x == 0) s_tot1 = t1;
    __syncthreads();
    const double mine = threadIdx.x == 0 ? t0 : threadIdx.x == 1 ? s_tot1 : 0.0;
    __syncthreads();
    const bool last = grid_reduce_finish_lanes<8, true>(mine, ro, scratch, gv);
    if (newton_step_and_wait(it, last, x, np, ro, ctl, host_out, host_flag, host_seq, &s_x, &s_status) != NEWTON_RUNNING) return;
  }
}

template <unsigned NT>
__global__ __launch_bounds__(256) void k_newton_s4(ModelView mv, ParamIdx params, NewtonParams np,
                                                   const double * sumtable,
                                                   const unsigned * ps, const unsigned * cs,
                                                   const unsigned * weights, const int * invariant,
                                                   unsigned N, unsigned R, ReduceOut ro,
                                                   NewtonControl * ctl, double * host_out,
                                                   unsigned long long * host_flag, unsigned long long host_seq)
{
  newton_loop_s4<NT>(mv, params, np, sumtable, ps, cs, weights, invariant, N, R, ro, ctl, host_out, host_flag, host_seq,
                     launch_grid());
}

// ---------------------------------------------------------------------------
// Several partitions under one branch length in ONE launch.  The loop over several partitions as separate launches --
// one per partition, each on its partition's stream -- needs a hardware queue per stream (the launches wait for one
// another on the device; the runtime has four queues unless the process asked for more before its first HIP call,
// and more of them slow the evaluations of many partitions down: pll_core.hip, pllhip_runtime_defaults).  Here every
// partition gets a contiguous run of the workgroups of one launch instead: its own scan grid (the one its blocking
// derivative call uses: same block totals, same order, same bits), its own tickets and totals, its family's loop;
// the runs meet after every scan in the first partition's control block exactly as the separate launches do
// (newton_step_and_wait).  Families: 4 states with the table in registers; 20 states, 33 .. 64 states and 2 .. 32 states
// streamed.
// grid = the sum of the partitions' scan grids (all co-resident), block = 256, dynamic LDS = the largest need.
// ---------------------------------------------------------------------------
constexpr unsigned NEWTON_KIND_S4 = 0, NEWTON_KIND_S20 = 1, NEWTON_KIND_S61 = 2, NEWTON_KIND_S61_RT = 3;
constexpr unsigned NEWTON_KIND_S16 = 4;          // + KS - 1: the 2 .. 32-state family with KS = ceil(S / 4) k-steps
struct NewtonMultiPart
{
  ModelView mv;
  ParamIdx params;
  const double * sumtable;
  const unsigned * ps, * cs, * weights;
  const int * invariant;
  ReduceOut ro;
  double xscale;
  unsigned N, nblk, R, rate_scalers;
  unsigned kind, first_block, nblocks, pad;
};
struct NewtonMultiArgs
{
  NewtonMultiPart part[NEWTON_MAX_PARTS];
  NewtonParams np;                  // (part / xscale are taken from the partition's entry)
  unsigned nparts, pad;
};

static_assert(sizeof(NewtonMultiArgs) <= 3072, "k_newton_multi takes its partitions by value");

// ALL: with the loops of the 2 .. 32-state family (eight more instantiations; the kernel without them -- the common
// mixes of DNA, protein and codon partitions -- keeps its registers: 256 without a spill against 74 spilled with them)
template <bool ALL>
__device__ inline void newton_multi_run(const NewtonMultiPart & P, unsigned p, const NewtonParams & np0, NewtonControl * ctl,
                                        double * host_out, unsigned long long * host_flag, unsigned long long host_seq)
{
  NewtonParams np = np0;
  np.part = p;
  np.xscale = P.xscale;
  if (p) np.stall_block = ~0u;
  const GridView gv{blockIdx.x - P.first_block, P.nblocks};
  switch (P.kind)
  {
    case NEWTON_KIND_S4:
      newton_loop_s4<2>(P.mv, P.params, np, P.sumtable, P.ps, P.cs, P.weights, P.invariant, P.N, P.R, P.ro, ctl, host_out,
                        host_flag, host_seq, gv);
      break;
    case NEWTON_KIND_S20:
      newton_loop<5, 20, 0>(P.mv, P.params, np, P.sumtable, P.ps, P.cs, P.weights, P.invariant, P.N, P.nblk, P.R, P.ro,
                            P.rate_scalers, ctl, host_out, host_flag, host_seq, gv);
      break;
    case NEWTON_KIND_S61:
      newton_loop<S61_KS, S61_S, 0>(P.mv, P.params, np, P.sumtable, P.ps, P.cs, P.weights, P.invariant, P.N, P.nblk, P.R, P.ro,
                                    P.rate_scalers, ctl, host_out, host_flag, host_seq, gv);
      break;
    case NEWTON_KIND_S61_RT:
      newton_loop<S61_KS, 0, 0>(P.mv, P.params, np, P.sumtable, P.ps, P.cs, P.weights, P.invariant, P.N, P.nblk, P.R, P.ro,
                                P.rate_scalers, ctl, host_out, host_flag, host_seq, gv);
      break;
#define PLLHIP_KIND_S16(KK)                                                                                              \
    case NEWTON_KIND_S16 + KK - 1:                                                                                       \
      if (ALL)                                                                                                           \
      newton_loop<KK, 0, 0>(P.mv, P.params, np, P.sumtable, P.ps, P.cs, P.weights, P.invariant, P.N, P.nblk, P.R, P.ro,   \
                            P.rate_scalers, ctl, host_out, host_flag, host_seq, gv);                                     \
      break;
    PLLHIP_KIND_S16(1) PLLHIP_KIND_S16(2) PLLHIP_KIND_S16(3) PLLHIP_KIND_S16(4)
    PLLHIP_KIND_S16(5) PLLHIP_KIND_S16(6) PLLHIP_KIND_S16(7) PLLHIP_KIND_S16(8)
#undef PLLHIP_KIND_S16
    default:
      break;
  }
}

// ALL: with the loops of the 2 .. 32-state family (eight more instantiations; the kernel without them -- the common
// mixes of DNA, protein and codon partitions -- keeps its registers: 256 without a spill against 74 spilled with them)
template <bool ALL>
__global__ __launch_bounds__(256, 2) void k_newton_multi(NewtonMultiArgs a, NewtonControl * ctl, double * host_out,
                                                         unsigned long long * host_flag, unsigned long long host_seq)
{
  // the partition of this workgroup (block-uniform)
  unsigned p = 0;
  for (unsigned k = 1; k < a.nparts; ++k) if (blockIdx.x >= a.part[k].first_block) p = k;
  if constexpr (!ALL)
    newton_multi_run<false>(a.part[p], p, a.np, ctl, host_out, host_flag, host_seq);     // (read straight from the arguments)
  else
  {
    // (with the eight loops more, that form made the compiler copy the whole argument block to scratch memory -- 2.4 KB
    // per thread --: here the entry is picked with constant indices)
    NewtonMultiPart picked = a.part[0];
#pragma unroll
    for (unsigned k = 1; k < NEWTON_MAX_PARTS; ++k) if (k == p) picked = a.part[k];
    newton_multi_run<true>(picked, p, a.np, ctl, host_out, host_flag, host_seq);
  }
}

} // namespace pllhip
