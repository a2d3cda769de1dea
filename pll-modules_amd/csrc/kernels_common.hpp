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

// Grid-wide sums that end where the caller wants them -- pinned, device-mapped host
// memory (a scalar-returning call: no device->host copy) or a device-resident slot (a
// deferred result, all-reduced by RCCL afterwards) -- in ONE launch and bit-reproducible
// run to run:
//   every block publishes its totals with write-through (sc1) stores, waits for them and
//   draws a ticket (two levels: 8 shard counters on lines of their own, then a top
//   counter, so that no single address takes more than gridDim.x / 8 adds); the block
//   that draws the last ticket acquires once and adds all block totals in a FIXED order
//   (by block index), whatever the arrival order was.  No release fence anywhere:
//   `__threadfence()` in every block serialises on the L2 write-back (measured 3x slower
//   in round 1; MI355X_MICROARCH.md, hand-off forms).
// fused == 0 keeps the two-launch form (block totals, then k_final_sum).
// Staging loops (global memory -> LDS / global): out(e, in(e)) for e = threadIdx.x, + blockDim.x, ... < n with U
// loads of a thread in flight.  The plain form "for (e = tid; e < n; e += blockDim.x) lds[e] = g[e]" compiles to
// load -> s_waitcnt vmcnt(0) -> ds_write per iteration: one memory round trip per 256 doubles, 16 in a row for the
// fragments of one 61-state matrix -- 20 us at the head of every workgroup, which is all a single operation on a
// 25 k-site partition takes otherwise.  in(e) must be a load that is valid for every e < n (no branch around it).
template <unsigned U, class In, class Out>
__device__ inline void staged_loop(unsigned n, In in, Out out)
{
  const unsigned T = blockDim.x;
  for (unsigned base = threadIdx.x; base < n; base += U * T)
  {
    decltype(in(0u)) v[U];
#pragma unroll
    for (unsigned u = 0; u < U; ++u) { const unsigned e = base + u * T; v[u] = in(e < n ? e : base); }
#pragma unroll
    for (unsigned u = 0; u < U; ++u) { const unsigned e = base + u * T; if (e < n) out(e, v[u]); }
  }
}

// dst[e] = src[e], e < n
template <unsigned U = 8>
__device__ inline void staged_copy(double * dst, const double * src, unsigned n)
{
  staged_loop<U>(n, [=](unsigned e) { return src[e]; }, [=](unsigned e, double v) { dst[e] = v; });
}

constexpr unsigned REDUCE_SHARDS = 8;
constexpr unsigned REDUCE_SHARD_STRIDE = 1024;   // unsigned words: 4 KiB apart
constexpr unsigned REDUCE_MAX_Q = 16;            // quantities per launch (8 trial lengths x {df, ddf})

struct ReduceOut
{
  double * block_out;          // [quantities][gridDim.x] device scratch
  unsigned * counter;          // [REDUCE_SHARDS * REDUCE_SHARD_STRIDE + 1], zero between launches
  double * dst;                // [quantities] mapped host memory or device slot
  unsigned long long * flag;   // or null: takes `seq` once dst is visible to the host
  unsigned long long seq;
  int fused;
  unsigned nq;                 // quantities to publish (a padded multi-length launch computes more)
};

inline ReduceOut reduce_out(const Engine * e)
{
  ReduceOut ro;
  ro.block_out = e->d_partials;
  ro.counter = e->d_counter;
  ro.dst = e->sink.dst;
  ro.flag = e->sink.flag;
  ro.seq = e->sink.seq;
  ro.fused = e->fused_finish ? 1 : 0;
  ro.nq = e->sink.nq;
  return ro;
}

// sum of block_out[q][0 .. nblocks) in the fixed order both finishing forms share:
// thread-strided partial sums, wave butterfly, four waves
template <bool SC1>
__device__ inline double final_sum_256(const double * col, unsigned nblocks, double * scratch)
{
  double a = 0.0;
  for (unsigned b = threadIdx.x; b < nblocks; b += 256)
    a += SC1 ? __hip_atomic_load(col + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : col[b];
  return block_sum_256(a, scratch);
}

__device__ inline void publish_result(const ReduceOut & ro)
{
  // values first, then the sequence word the host polls
  if (ro.flag)
  {
    __threadfence_system();
    __hip_atomic_store(ro.flag, ro.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// what every block does after its totals are stored (write-through) and waited for:
// draw the ticket(s); the last block adds all block totals and publishes.
//
// Memory-model note.  The hand-off is: producer -- relaxed agent-scope atomic STORES of its totals,
// `s_waitcnt vmcnt(0)` (the stores have been acknowledged), relaxed agent-scope fetch-add on the ticket;
// consumer (the block that draws the last ticket) -- acquire fence at agent scope, relaxed agent-scope atomic
// LOADS of all totals.  In the language's model a relaxed store followed by a relaxed RMW gives no
// happens-before edge; the code relies on what the AMDGPU backend documents for gfx942 / gfx950 in the LLVM
// "User Guide for AMDGPU Backend", section "Memory Model" (tables "AMDHSA Memory Model Code Sequences
// GFX942"): an agent-scope atomic store is emitted with sc1 = 1, i.e. written through to the memory every XCD's
// L2 is coherent with; `s_waitcnt vmcnt(0)` is exactly what a release at agent scope inserts in front of
// the following atomic to order earlier stores; the acquire fence invalidates the consumer's L1 / non-coherent
// L2 lines (buffer_inv sc1), and agent-scope atomic loads bypass them anyway.  So the instruction sequence IS
// the release-acquire sequence of that guide, minus the `buffer_wbl2 sc1` write-back of the whole L2 that a
// __threadfence() / release fence adds -- which is what made the straightforward form 3x slower (thousands of
// short streaming blocks each flushing their XCD's L2).  It is outside the C++ model, therefore guarded:
// tests/test_gpu_results.py::test_single_launch_reduction_equals_two_launch_form compares it bit for bit with
// the two-launch form (PLLHIP_FUSED_FINISH=0), which needs no such argument, and a port to another target
// must re-derive it (or set fused_finish = false).
// LOOP: the launch goes on after the sum (k_newton_mfma): the caller resets the tickets and tells the other
// blocks itself; returns whether this block drew the last ticket (thread 0 then finds the totals in ro.dst)
// A kernel that runs the scans of several partitions side by side (k_newton_multi, kernels_newton_s4.hpp) gives every
// partition a contiguous run of its workgroups: what a scan and its reduction take for "the grid" is then that run
// -- workgroup b of G -- and not the launch.  Everywhere else it is the launch (the default argument).
struct GridView { unsigned b, G; };
__device__ inline GridView launch_grid() { return GridView{blockIdx.x, gridDim.x}; }

template <int Q, bool LOOP = false>
__device__ inline bool grid_reduce_tail(const ReduceOut & ro, double * scratch, GridView gv = launch_grid())
{
  const unsigned G = gv.G, b = gv.b;
  unsigned * s_last = reinterpret_cast<unsigned *>(scratch);   // block_sum_256 left it free
  if (threadIdx.x == 0)
  {
    unsigned last = 0;
    const unsigned shard = b % REDUCE_SHARDS;
    const unsigned in_shard = (G - shard + REDUCE_SHARDS - 1) / REDUCE_SHARDS;     // blocks with this b % 8
    const unsigned t1 = __hip_atomic_fetch_add(ro.counter + shard * REDUCE_SHARD_STRIDE, 1u,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t1 == in_shard - 1)
    {
      const unsigned shards = G < REDUCE_SHARDS ? G : REDUCE_SHARDS;
      const unsigned t2 = __hip_atomic_fetch_add(ro.counter + REDUCE_SHARDS * REDUCE_SHARD_STRIDE, 1u,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t2 == shards - 1);
    }
    *s_last = last;
  }
  __syncthreads();
  const unsigned last = *s_last;
  __syncthreads();
  if (!last) return false;                // block-uniform
  if (threadIdx.x == 0)
  {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // all quantities at once: the loads of every quantity are in flight together (one memory
  // latency, not Q of them -- the totals were stored write-through, so they come from memory),
  // each quantity still adds its blocks in ascending order per thread
  double a[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) a[q] = 0.0;
  for (unsigned bb = threadIdx.x; bb < G; bb += 256)
  {
    double v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
      v[q] = ((unsigned)q < ro.nq)
                 ? __hip_atomic_load(ro.block_out + (size_t)q * G + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                 : 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) a[q] += v[q];
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    if ((unsigned)q >= ro.nq) break;
    const double t = block_sum_256(a[q], scratch);
    if (threadIdx.x == 0) ro.dst[q] = t;
  }
  if (LOOP) return true;
  if (threadIdx.x <= REDUCE_SHARDS)       // tickets back to zero for the next launch on this stream
    __hip_atomic_store(ro.counter + threadIdx.x * REDUCE_SHARD_STRIDE, 0u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) publish_result(ro);
  return true;
}

// block totals in thread 0 (tot[0 .. Q))
template <int Q>
__device__ inline void grid_reduce_finish(const double (&tot)[Q], const ReduceOut & ro,
                                          double * scratch)
{
  const unsigned G = gridDim.x, b = blockIdx.x;
  if (threadIdx.x == 0)
  {
    if (!ro.fused)
    {
#pragma unroll
      for (int q = 0; q < Q; ++q) ro.block_out[(size_t)q * G + b] = tot[q];
    }
    else
    {
#pragma unroll
      for (int q = 0; q < Q; ++q)
        __hip_atomic_store(ro.block_out + (size_t)q * G + b, tot[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (ro.fused) grid_reduce_tail<Q>(ro, scratch);
}

// block total of quantity t in thread t (t < Q <= 64: all of them lanes of wave 0, whose one
// wait covers every store)
template <int Q, bool LOOP = false>
__device__ inline bool grid_reduce_finish_lanes(double mine, const ReduceOut & ro, double * scratch, GridView gv = launch_grid())
{
  const unsigned G = gv.G, b = gv.b;
  if (threadIdx.x < 64)
  {
    if (threadIdx.x < Q)
    {
      if (!ro.fused) ro.block_out[(size_t)threadIdx.x * G + b] = mine;
      else __hip_atomic_store(ro.block_out + (size_t)threadIdx.x * G + b, mine, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ro.fused) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (ro.fused) return grid_reduce_tail<Q, LOOP>(ro, scratch, gv);
  return false;
}

// two-launch form: block_out[q][nblocks] -> dst[q]; one block of 256 threads
__global__ __launch_bounds__(256) void k_final_sum(ReduceOut ro, unsigned nblocks, unsigned nq)
{
  __shared__ double scratch[4];
  for (unsigned q = 0; q < nq && q < ro.nq; ++q)
  {
    const double t = final_sum_256<false>(ro.block_out + (size_t)q * nblocks, nblocks, scratch);
    if (threadIdx.x == 0) ro.dst[q] = t;
  }
  if (threadIdx.x == 0) publish_result(ro);
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

// PLL_ATTRIB_RATE_SCALERS: scaler[n*R + r].  A site's terms are brought to the smallest count
// among its rates; a rate d counts above it is multiplied by 2^(-256 min(d, 4)) (libpll-2's
// PLL_SCALE_RATE_MAXDIFF; same rule as the oracle, oracle/orc_kernels.c rate_counts)
constexpr unsigned SCALE_RATE_MAXDIFF = 4;

__device__ inline unsigned rate_min_count(const unsigned * ps, const unsigned * cs, unsigned long long n, unsigned R)
{
  unsigned mn = ~0u;
  for (unsigned r = 0; r < R; ++r)
  {
    const unsigned c = (ps ? ps[n * R + r] : 0u) + (cs ? cs[n * R + r] : 0u);
    mn = c < mn ? c : mn;
  }
  return mn;
}

__device__ inline double rate_factor(const unsigned * ps, const unsigned * cs, unsigned long long n, unsigned R,
                                     unsigned r, unsigned min_cnt)
{
  unsigned d = (ps ? ps[n * R + r] : 0u) + (cs ? cs[n * R + r] : 0u) - min_cnt;
  d = d > SCALE_RATE_MAXDIFF ? SCALE_RATE_MAXDIFF : d;
  return d ? ldexp(1.0, -256 * (int)d) : 1.0;
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

// ---------------------------------------------------------------------------
// device-resident schedules (PlanOp / PlanChain, engine.h): entries are fetched through the
// constant address space, i.e. with scalar loads into SGPRs -- the index is wave-uniform and the
// buffer is written by a copy that precedes the launch
// ---------------------------------------------------------------------------
template <class T>
__device__ inline T plan_fetch(const T * p)
{
  static_assert(sizeof(T) % 8 == 0, "plan entries are whole 64-bit words");
  constexpr unsigned W = sizeof(T) / 8;
  typedef const unsigned long long __attribute__((address_space(4))) * const_words;
  const_words c = (const_words)(unsigned long long)p;
  unsigned long long w[W];
#pragma unroll
  for (unsigned k = 0; k < W; ++k) w[k] = c[k];
  T out;
  __builtin_memcpy(&out, w, sizeof(T));
  return out;
}

// A pointer that came out of a fetched plan entry has lost its address space (the compiler would
// use flat loads / stores, which also count against LDS traffic): say that it is global memory.
template <class T>
__device__ inline T * as_global(T * p)
{
  typedef __attribute__((address_space(1))) T * global_ptr;
  return (T *)(global_ptr)(unsigned long long)p;
}

// where the operations and chains of a schedule are: in device memory, or in the kernel arguments themselves (PlanView is
// the FIRST argument of every traversal kernel: its bytes start the kernel-argument segment, which is ordinary memory)
__device__ inline void plan_bases(const PlanView & plan, const PlanOp *& ops, const PlanChain *& chains)
{
  if (plan.inline_ops)
  {
    const unsigned long long base = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PlanView, inl);
    ops = reinterpret_cast<const PlanOp *>(base);
    chains = reinterpret_cast<const PlanChain *>(base + plan.inline_ops);
  }
  else { ops = plan.ops; chains = plan.chains; }
}

__device__ inline PlanOp plan_fetch_op(const PlanOp * p)
{
  PlanOp po = plan_fetch(p);
  OpDesc & d = po.d;
  d.clv1 = as_global(d.clv1);       d.codes1 = as_global(d.codes1);
  d.pmat1 = as_global(d.pmat1);     d.lut1 = as_global(d.lut1);
  d.clv2 = as_global(d.clv2);       d.codes2 = as_global(d.codes2);
  d.pmat2 = as_global(d.pmat2);     d.lut2 = as_global(d.lut2);
  d.scaler1 = as_global(d.scaler1); d.scaler2 = as_global(d.scaler2);
  d.parent = as_global(d.parent);   d.parent_scaler = as_global(d.parent_scaler);
  d.pfrag1 = as_global(d.pfrag1);   d.pfrag2 = as_global(d.pfrag2);
  return po;
}

// Workgroups per grid row of a launch whose rows are the chains of a round.  Every workgroup of a row stages
// the tables of its chain before its waves stream site blocks, so a row should have no more workgroups than
// it takes to fill the chip TOGETHER with the other rows: with many rows (a round of a bushy tree, or the
// rounds of many partitions in one launch, pllhip_update_partials_batch) a row gets few workgroups and each
// of them many blocks, and the staging is paid once per row instead of once per eight blocks.
// PLLHIP_ROUND_WGS: workgroups per CU the launch aims for (0: a row takes what its partition could use alone).
// The eigen-basis operands of the sumtable kernels (k_sumtable_prep_*: pi V and V^-1 per rate, and their tip
// tables) depend on the model, the parameter indices and the code table, not on the branch: true when the scratch
// area has to be prepared (again) -- once per model change instead of once per sumtable (a branch-length pass at
// 125 k protein sites: 397 launches of 20 us; an SPR round at 61 states: 1 372 of 36 us)
static bool sum_prep_needed(Engine * e, const ParamIdx & params, bool want_lut)
{
  bool same = e->sum_prep_model == e->model_generation && e->sum_prep_lut_codes == e->lut_codes &&
              e->sum_prep_tipmap == e->tipmap_codes_uploaded && (e->sum_prep_has_lut || !want_lut);
  for (unsigned r = 0; same && r < e->R && r < 16; ++r) same = e->sum_prep_params[r] == params.v[r];
  if (same) return false;
  e->sum_prep_model = e->model_generation;
  e->sum_prep_lut_codes = e->lut_codes;
  e->sum_prep_tipmap = e->tipmap_codes_uploaded;
  e->sum_prep_has_lut = want_lut;
  for (unsigned r = 0; r < e->R && r < 16; ++r) e->sum_prep_params[r] = params.v[r];
  return true;
}

// keep: workgroups a row keeps whatever the other rows take (the ones it needs to share a LARGE partition out
// finely enough -- few long-lived workgroups per row leave a tail at the end of every round)
static unsigned round_grid(const Engine * e, unsigned gx, unsigned rows, unsigned per_cu_default = 4u, unsigned keep = 0u)
{
  static const int env = getenv("PLLHIP_ROUND_WGS") ? atoi(getenv("PLLHIP_ROUND_WGS")) : -1;
  const unsigned per_cu = (env >= 0 && per_cu_default >= 4u) ? (unsigned)env : per_cu_default;
  if (rows <= 1 || per_cu == 0) return gx;
  const unsigned share = (e->cu_count * per_cu + rows - 1) / rows;
  return std::max(1u, std::min(gx, std::max(share, keep)));
}

} // namespace pllhip
