// kernels_common.hpp -- device helpers shared by every kernel family:
// deterministic block reductions (no float atomics: SPR rounds assert
// run-to-run reproducibility, src/algorithm/algo_search.c:1453-1457) and the
// per-site log-likelihood assembly.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include "engine.h"

namespace pllhip {

constexpr double SCALE_FACTOR = 115792089237316195423570985008687907853269984665640564039457584007913129639936.0; // 2^256
constexpr double SCALE_THRESHOLD = 1.0 / SCALE_FACTOR;
constexpr double LN_SCALE = 177.445678223345993274; // 256 ln 2

// model block offsets, see Engine::d_model
struct ModelView
{
  const double * base;
  unsigned off_rates, off_weights, off_pinv, off_freqs, off_evals, off_evecs, off_ievecs;
  unsigned S, Sp;
  __device__ const double * rates() const { return base + off_rates; }
  __device__ const double * weights() const { return base + off_weights; }
  __device__ const double * pinv() const { return base + off_pinv; }
  __device__ const double * freqs(unsigned m) const { return base + off_freqs + (size_t)m * Sp; }
  __device__ const double * evals(unsigned m) const { return base + off_evals + (size_t)m * Sp; }
  __device__ const double * evecs(unsigned m) const { return base + off_evecs + (size_t)m * S * Sp; }
  __device__ const double * ievecs(unsigned m) const { return base + off_ievecs + (size_t)m * S * Sp; }
};

typedef ParamIdxHost ParamIdx;

// a node's conditional likelihoods: full CLV or 1-byte tip codes
struct NodeRef
{
  const double * clv;
  const uint8_t * codes;
};

// wave-level sum by butterfly shuffles: every lane ends with the same total,
// and the association order is fixed by the lane id, hence deterministic
__device__ inline double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block sum of `v` over 256 threads; result valid in thread 0.
// scratch: 4 doubles of LDS per reduced quantity.
__device__ inline double block_sum_256(double v, double * scratch)
{
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return t;
}

// Grid-wide sums that end in host-visible memory without a device->host copy:
// every block stores its total; a second, single-block kernel (k_final_sum, the
// kernel boundary gives the visibility) adds the block totals in a FIXED order
// and writes the result into pinned, device-mapped host memory.  Bit-reproducible
// run to run.  (An in-kernel "last block finishes" variant with agent-scope
// fences was measured 3x slower here: thousands of release fences from short
// streaming blocks serialise on the XCD L2 write-back.)
struct ReduceOut
{
  double * block_out;     // [quantities][gridDim.x] device scratch
  unsigned * counter;     // zero before the launch; reset by the last block
  double * host_result;   // device pointer of pinned host memory, [quantities]
};

inline ReduceOut reduce_out(const Engine * e)
{
  ReduceOut ro;
  ro.block_out = e->d_partials;
  ro.counter = e->d_counter;
  ro.host_result = e->d_result;
  return ro;
}

template <int Q>
__device__ inline void grid_reduce_finish(const double (&tot)[Q], const ReduceOut & ro,
                                          double * scratch)
{
  (void)scratch;
  if (threadIdx.x == 0)
  {
#pragma unroll
    for (int q = 0; q < Q; ++q) ro.block_out[(size_t)q * gridDim.x + blockIdx.x] = tot[q];
  }
}

// block_out[q][nblocks] -> host_result[q]; one block of 256 threads
// host_result[RESULT_SEQ_SLOT] takes `seq` (as an integer) once the sums are visible to the
// host: the caller may poll it instead of paying a stream synchronisation.
constexpr unsigned RESULT_SEQ_SLOT = 7;

__global__ __launch_bounds__(256) void k_final_sum(const double * block_out, unsigned nblocks,
                                                   unsigned nq, double * host_result,
                                                   unsigned long long seq)
{
  __shared__ double scratch[4];
  for (unsigned q = 0; q < nq; ++q)
  {
    double a = 0.0;
    for (unsigned b = threadIdx.x; b < nblocks; b += 256) a += block_out[(size_t)q * nblocks + b];
    const double t = block_sum_256(a, scratch);
    if (threadIdx.x == 0) host_result[q] = t;
  }
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_result) + RESULT_SEQ_SLOT, seq,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ inline void grid_reduce_finish1(double a, const ReduceOut & ro, double * scratch)
{
  const double t[1] = {a};
  grid_reduce_finish<1>(t, ro, scratch);
}

__device__ inline void grid_reduce_finish2(double a, double b, const ReduceOut & ro, double * scratch)
{
  const double t[2] = {a, b};
  grid_reduce_finish<2>(t, ro, scratch);
}

// per-site side values of a reduction kernel (scaler counts of both ends, pattern weight)
// for the two sites of a lane: fetched at the top of a block iteration -- they are needed
// only after the reduction over states and rates, and a load issued there would sit in
// front of the log / division with its full latency
struct SiteSide { unsigned cnt_e, cnt_o, w_e, w_o; };

__device__ inline SiteSide load_site_side(const unsigned * ps, const unsigned * cs, const unsigned * weights,
                                          size_t site0, unsigned N, bool active)
{
  SiteSide sd = {0u, 0u, 0u, 0u};
  if (active)
  {
    if (site0 < N)
    {
      sd.cnt_e = (ps ? ps[site0] : 0u) + (cs ? cs[site0] : 0u);
      sd.w_e = weights[site0];
    }
    if (site0 + 1 < N)
    {
      sd.cnt_o = (ps ? ps[site0 + 1] : 0u) + (cs ? cs[site0 + 1] : 0u);
      sd.w_o = weights[site0 + 1];
    }
  }
  return sd;
}

// log of the site likelihood.  x is the likelihood carrying `cnt` scaling
// steps (true value x * 2^(-256 cnt)); inv is the unscaled invariant-site
// term.  Same case split as the oracle (oracle/orc_kernels.c site_loglh).
__device__ inline double site_loglh(double x, unsigned cnt, double inv)
{
  if (inv > 0.0 && cnt > 0)
  {
    const double xt = (cnt <= 3) ? ldexp(x, -256 * (int)cnt) : 0.0;
    return log(xt + inv);
  }
  return log(x + inv) - (double)cnt * LN_SCALE;
}

} // namespace pllhip
