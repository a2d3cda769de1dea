// pll_core.hip -- the hot path of include/pll.h (tier B0) on gfx950, plus the
// engine services behind it and the pllhip_* extension API.
//
// Entry points and the reference call sites they serve:
//   pll_update_prob_matrices           src/tree/treeinfo.c:854 (count = 1 per branch!)
//   pll_update_partials                src/tree/treeinfo.c:1037,
//                                      src/optimize/pll_optimize.c:748-775 (count = 1)
//   pll_compute_edge_loglikelihood     src/tree/treeinfo.c:1049
//   pll_compute_root_loglikelihood     src/optimize/pll_optimize.c:329
//   pll_update_sumtable                src/optimize/pll_optimize.c:800, 1468
//   pll_compute_likelihood_derivatives src/optimize/pll_optimize.c:307, 1151, 1249
//
// Execution model: one HIP stream per partition; every call enqueues kernels
// and returns, except the calls that return a scalar (lnL, derivatives), which
// copy <= 16 KB of per-block partial sums to pinned host memory, synchronise
// and finish the sum on the host in a fixed order (bit-reproducible).
// Operation lists are level-scheduled: ops of one dependency level share one
// launch (grid.y = op), their descriptors travel by value in the kernel
// arguments, so no host->device copy sits on the critical path.
#include "engine.h"
#include "kernels_common.hpp"
#include "kernels_generic.hpp"
#include "kernels_s4.hpp"
#include "kernels_s20.hpp"
#include "kernels_s61.hpp"
#include "kernels_s16.hpp"
#include "kernels_repeats.hpp"
#include "kernels_newton_s4.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <time.h>

namespace pllhip {

static thread_local int g_device = -1;

bool hip_ok(hipError_t e, const char * what)
{
  if (e == hipSuccess) return true;
  set_error(PLL_ERROR_HIP_RUNTIME, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return false;
}

static int current_device()
{
  if (g_device < 0)
  {
    const char * env = getenv("PLLHIP_DEVICE");
    g_device = env ? atoi(env) : 0;
  }
  return g_device;
}

int flush_pmatrices(pll_partition_t * p);
static int ensure_luts(pll_partition_t * p);
static int update_partials_impl(pll_partition_t * p, const pll_operation_t * ops, unsigned count);
static int transient_materialize(Engine * e, const std::vector<unsigned> & want);
static int transient_flush_all(Engine * e);
static bool cherry_storage(Engine * e, unsigned node, unsigned nclasses);
// tip lookup tables exist: PLL_ATTRIB_PATTERN_TIP, or the engine's own codes (Engine::shadow_codes)
static bool lut_active(const Engine * e) { return e->coded_tips || e->shadow_codes; }
static unsigned codes_in_use(const Engine * e, const pll_partition_t * p)
{
  return e->shadow_codes ? (unsigned)e->shadow_tipmap.size() : p->maxstates;
}
// the operations above tip `idx` read it through byte codes and lookup tables
static bool tip_coded(const Engine * e, unsigned idx)
{
  return idx < e->tips && (e->coded_tips || (e->shadow_codes && e->tip_has_codes[idx]));
}
static void batch_free(BatchPlan * b);
static void batch_free_hook(BatchPlan * b) { batch_free(b); }

// ---------------------------------------------------------------------------
// engine-internal sharding: which devices the partitions created next are spread over
// ---------------------------------------------------------------------------
static thread_local std::vector<int> g_shard_devices;
static thread_local bool g_shard_devices_set = false;
static thread_local bool g_building_shards = false;

static std::vector<int> shard_devices()
{
  if (!g_shard_devices_set)
  {
    g_shard_devices_set = true;
    if (const char * env = getenv("PLLHIP_SHARD_DEVICES"))
      for (const char * c = env; *c; )
      {
        char * end = nullptr;
        const long v = strtol(c, &end, 10);
        if (end == c) break;
        g_shard_devices.push_back((int)v);
        c = (*end == ',') ? end + 1 : end;
      }
  }
  return g_shard_devices;
}

static ModelView model_view(const Engine * e)
{
  ModelView mv;
  mv.base = e->d_model;
  mv.off_rates = (unsigned)e->off_rates;
  mv.off_weights = (unsigned)e->off_weights;
  mv.off_pinv = (unsigned)e->off_pinv;
  mv.off_freqs = (unsigned)e->off_freqs;
  mv.off_evals = (unsigned)e->off_evals;
  mv.off_evecs = (unsigned)e->off_evecs;
  mv.off_ievecs = (unsigned)e->off_ievecs;
  mv.S = e->S;
  mv.Sp = e->Sp;
  return mv;
}

template <typename T>
static bool dev_alloc(T ** ptr, size_t count, const char * what)
{
  *ptr = nullptr;
  if (!count) count = 1;
  hipError_t err = hipMalloc(reinterpret_cast<void **>(ptr), count * sizeof(T));
  if (err != hipSuccess)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "hipMalloc of %zu bytes for %s failed: %s",
              count * sizeof(T), what, hipGetErrorString(err));
    *ptr = nullptr;
    return false;
  }
  return true;
}

// Kernel arguments in device memory (the ROCm runtime's HIP_FORCE_DEV_KERNARG): by default they sit in
// host memory, and the first wave of every workgroup pays a trip over the fabric for its operation
// descriptors -- 3 to 5 us on a kernel that runs for 10 (an SPR insertion's partial traversal, a
// 25 k-site codon slice: W3 -3 ... -6 %, C5 at 25 k sites 1.73 -> 1.65 ms per step).  Set when the
// library is loaded, i.e. before its first HIP call, unless the user has decided otherwise; a runtime
// that another component initialised earlier keeps its setting.
// The setting is process-global and setenv() is not safe against a concurrent getenv() in another thread of a
// host application that dlopen()s this library late: such an application sets HIP_FORCE_DEV_KERNARG itself
// before it starts threads (what bench.py and the tools do) and switches this off -- at run time with
// PLLHIP_SET_RUNTIME_DEFAULTS=0, or at build time with -DPLLHIP_NO_RUNTIME_DEFAULTS.
#ifndef PLLHIP_NO_RUNTIME_DEFAULTS
__attribute__((constructor)) static void pllhip_runtime_defaults()
{
  const char * opt = getenv("PLLHIP_SET_RUNTIME_DEFAULTS");
  if (opt && !atoi(opt)) return;
  if (!getenv("HIP_FORCE_DEV_KERNARG")) setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  // GPU_MAX_HW_QUEUES is left alone.  The runtime multiplexes a process's streams onto that many hardware queues (four
  // by default), in order within a queue; sixteen would give every partition stream its own -- which the Newton-Raphson
  // loop over several partitions needs (newton_multi_enabled) -- but cost the many-stream workloads a third of their
  // rate on one box: 64 DNA partitions of 10 k sites 3.6 -> 5.7 ms per evaluation, 32 protein partitions 5.6 -> 6.9,
  // 16 x 60 k DNA 3.3 -> 4.0 (tools/gpu_r4_queues.sh; between the two, 6 / 8 / 12 queues cost the same).
}
#endif

// partitions alive per device (whole-traversal launches are the default for a partition that has
// its device to itself)
static std::atomic<int> engines_on_device[64];

Engine * engine_create(pll_partition_t * p)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
  {
    set_error(PLL_ERROR_HIP_NODEVICE,
              "No HIP device available: the likelihood engine has no CPU fallback");
    return nullptr;
  }
  const int dev = current_device();
  if (dev >= ndev)
  {
    set_error(PLL_ERROR_HIP_NODEVICE, "HIP device %d requested, %d visible", dev, ndev);
    return nullptr;
  }
  // spread over several devices?  (not for the shards themselves, tiny partitions, or AB partitions:
  // their correction needs the constant patterns and the total pattern weight in one place)
  const std::vector<int> devs = g_building_shards ? std::vector<int>() : shard_devices();
  if (devs.size() >= 2 && !p->asc_bias_alloc && p->sites >= 64u * devs.size())
  {
    Engine * r = new (std::nothrow) Engine();
    if (!r) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate engine state"); return nullptr; }
    r->device = dev;
    r->S = p->states; r->Sp = p->states_padded; r->R = p->rate_cats; r->N = r->Nreal = p->sites;
    r->tips = p->tips; r->nodes = p->nodes; r->nscalers = p->scale_buffers;
    r->nmat = p->prob_matrices; r->nrm = p->rate_matrices;
    r->coded_tips = (p->attributes & PLL_ATTRIB_PATTERN_TIP) != 0;
    r->rate_scalers = (p->attributes & PLL_ATTRIB_RATE_SCALERS) != 0;
    const unsigned K = (unsigned)devs.size();
    const unsigned chunk = ((p->sites + K - 1) / K + S20_BS - 1) / S20_BS * S20_BS;   // whole 32-site blocks
    const int saved = g_device;
    g_building_shards = true;
    bool ok = true;
    for (unsigned k = 0; ok && k < K; ++k)
    {
      const unsigned first = std::min(p->sites, k * chunk), last = std::min(p->sites, (k + 1) * chunk);
      if (first == last) break;
      if (devs[k] < 0 || devs[k] % ndev != devs[k]) { /* wrap: lets a test shard over one visible device */ }
      g_device = ((devs[k] % ndev) + ndev) % ndev;
      pll_partition_t * c = pll_partition_create(p->tips, p->clv_buffers, p->states, last - first, p->rate_matrices,
                                                 p->prob_matrices, p->rate_cats, p->scale_buffers,
                                                 p->attributes & ~PLLHIP_ATTRIB_HOST_MIRRORS);
      ok = c != nullptr;
      if (ok) { r->shards.push_back(c); r->shard_first.push_back(first); }
    }
    g_building_shards = false;
    g_device = saved;
    if (!ok || r->shards.size() < 2)
    {
      for (pll_partition_t * c : r->shards) pll_partition_destroy(c);
      delete r;
      if (!ok) return nullptr;
    }
    else
    {
      r->shard_first.push_back(p->sites);
      r->family = engine_of(r->shards[0])->family;
      p->engine = r;
      return r;
    }
  }
  if (!hip_ok(hipSetDevice(dev), "hipSetDevice")) return nullptr;

  Engine * e = new (std::nothrow) Engine();
  if (!e)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate engine state");
    return nullptr;
  }
  e->device = dev;
  e->S = p->states; e->Sp = p->states_padded; e->R = p->rate_cats;
  e->Nreal = p->sites;
  e->N = p->sites + (p->asc_bias_alloc ? (unsigned)p->asc_additional_sites : 0u);
  e->tips = p->tips; e->nodes = p->nodes; e->nscalers = p->scale_buffers;
  e->nmat = p->prob_matrices; e->nrm = p->rate_matrices;
  e->coded_tips = (p->attributes & PLL_ATTRIB_PATTERN_TIP) != 0;
  e->tip_has_codes.assign(p->tips, 0);
  e->rate_scalers = (p->attributes & PLL_ATTRIB_RATE_SCALERS) != 0;
  // PLLHIP_SITE_REPEATS=0: never; =2: as if every partition had the attribute (the whole test suite under site repeats)
  const int env_repeats = getenv("PLLHIP_SITE_REPEATS") ? atoi(getenv("PLLHIP_SITE_REPEATS")) : 1;
  // (without PLL_ATTRIB_PATTERN_TIP -- libpll's own combination: the two attributes exclude one another there -- the
  // tips given through pll_set_tip_states are class nodes from the start: upload_tip_classes)
  // (ascertainment-bias columns are sites like any other for the class maps: weight 0, a class of their own each)
  e->site_repeats = ((p->attributes & PLL_ATTRIB_SITE_REPEATS) != 0 || env_repeats == 2) && !e->rate_scalers &&
                    env_repeats != 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) e->cu_count = prop.multiProcessorCount;

  const char * force = getenv("PLLHIP_FORCE_GENERIC");
  static const int no_s16 = getenv("PLLHIP_NO_S16") ? atoi(getenv("PLLHIP_NO_S16")) : 0;
  if (force && atoi(force)) e->family = KernelFamily::Generic;
  // per-rate scalers: every matrix-core family carries them (4 states go to the 2..16-state
  // family: the VALU 4-state kernels vote per site)
  else if (e->rate_scalers && e->S <= 16) e->family = no_s16 ? KernelFamily::Generic : KernelFamily::S16;
  else if (e->S == 4 && (e->R & (e->R - 1)) == 0) e->family = KernelFamily::S4;
  else if (e->S == 20 && e->R <= 8) e->family = KernelFamily::S20;
  // 33 .. 64 states: the codon family (61 states: its own instantiation with the state count at compile time)
  else if (e->S >= 33 && e->S <= 64) e->family = KernelFamily::S61;
  // 2 .. 32 states (17 .. 32: two M tiles): every alphabet the 4- and 20-state families do not take
  else if (s16_supported(e->S, e->R) && !no_s16) e->family = KernelFamily::S16;
  else e->family = KernelFamily::Generic;
  e->blocked = (e->family == KernelFamily::S20 || e->family == KernelFamily::S61 || e->family == KernelFamily::S16);
  // (PLLHIP_TIP_CLASSES=0: tips of partitions without pattern tips as vectors only)
  e->shadow_codes = e->family == KernelFamily::S61 && !e->coded_tips && p->asc_bias_alloc == 0 &&
                    !(getenv("PLLHIP_TIP_CLASSES") && atoi(getenv("PLLHIP_TIP_CLASSES")) == 0);
  e->rows = !e->blocked ? 0u : (e->family == KernelFamily::S16) ? 4u * ((e->S + 3u) / 4u)
                               : (e->family == KernelFamily::S61) ? S61_SP : e->Sp;
  e->nblk = (e->N + S20_BS - 1) / S20_BS;
  e->Nalloc = e->blocked ? e->nblk * S20_BS : e->N;
  e->sc_len = (size_t)e->Nalloc * (e->rate_scalers ? e->R : 1);

  bool ok = hip_ok(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate");
  if (ok) { e->counted = true; ++engines_on_device[e->device & 63]; }
  const size_t clv_len = e->blocked ? (size_t)e->nblk * e->R * e->rows * S20_BS : (size_t)e->N * e->R * e->Sp;
  e->clv_len = clv_len;
  e->d_clv.assign(e->nodes, nullptr);
  e->d_codes.assign(e->tips, nullptr);
  for (unsigned i = 0; ok && i < e->nodes; ++i)
  {
    // (33 .. 64 states without pattern tips: codes of the engine's own next to the vector)
    if (i < e->tips && e->shadow_codes)
      ok = dev_alloc(&e->d_codes[i], (size_t)e->Nalloc, "tip codes") &&
           hip_ok(hipMemsetAsync(e->d_codes[i], 0, e->Nalloc ? e->Nalloc : 1, e->stream), "memset codes");
    if (i < e->tips && e->coded_tips)
      ok = dev_alloc(&e->d_codes[i], (size_t)e->Nalloc, "tip codes") &&
           hip_ok(hipMemsetAsync(e->d_codes[i], 0, e->Nalloc ? e->Nalloc : 1, e->stream), "memset codes");
    else
      ok = dev_alloc(&e->d_clv[i], clv_len, "CLV") &&
           hip_ok(hipMemsetAsync(e->d_clv[i], 0, (clv_len ? clv_len : 1) * sizeof(double), e->stream),
                  "memset CLV");
  }
  const size_t pm_len = (size_t)e->nmat * e->R * e->S * e->Sp;
  ok = ok && dev_alloc(&e->d_scalers, (size_t)e->nscalers * e->sc_len, "scalers");
  ok = ok && hip_ok(hipMemsetAsync(e->d_scalers, 0,
                                   std::max<size_t>(1, (size_t)e->nscalers * e->sc_len) * sizeof(unsigned),
                                   e->stream), "memset scalers");
  ok = ok && dev_alloc(&e->d_pmat, pm_len, "P-matrices");
  if (e->family == KernelFamily::S20)
    ok = ok && dev_alloc(&e->d_pfrag, std::max<size_t>(1, (size_t)e->nmat * e->R * 400), "P-matrix fragments") &&
         hip_ok(hipMemsetAsync(e->d_pfrag, 0, std::max<size_t>(1, (size_t)e->nmat * e->R * 400) * sizeof(double), e->stream),
                "memset fragments");
  ok = ok && hip_ok(hipMemsetAsync(e->d_pmat, 0, std::max<size_t>(1, pm_len) * sizeof(double), e->stream),
                    "memset pmat");
  ok = ok && dev_alloc(&e->d_weights, (size_t)e->N, "pattern weights");
  ok = ok && dev_alloc(&e->d_tipmap, (size_t)PLL_ASCII_SIZE, "tipmap");
  ok = ok && hip_ok(hipMemsetAsync(e->d_tipmap, 0, PLL_ASCII_SIZE * sizeof(unsigned long long), e->stream),
                    "memset tipmap");
  ok = ok && dev_alloc(&e->d_partials, (size_t)REDUCE_QUANTITIES * REDUCE_BLOCKS, "reduction partials");
  ok = ok && dev_alloc(&e->d_counter, (size_t)REDUCE_COUNTER_WORDS, "reduction tickets");
  ok = ok && hip_ok(hipMemsetAsync(e->d_counter, 0, REDUCE_COUNTER_WORDS * sizeof(unsigned), e->stream),
                    "memset tickets");
  ok = ok && hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_result), RESULT_WORDS * sizeof(double),
                                  hipHostMallocMapped), "hipHostMalloc result");
  if (ok) memset(e->h_result, 0, RESULT_WORDS * sizeof(double));   // the sequence word starts at 0 (recycled pinned pages)
  ok = ok && hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&e->d_result), e->h_result, 0),
                    "hipHostGetDevicePointer");
  if (const char * ff = getenv("PLLHIP_FUSED_FINISH")) e->fused_finish = atoi(ff) != 0;
  if (p->asc_bias_alloc)
  {
    const size_t bytes = sizeof(double) * MAX_TRIAL_LENGTHS * 64 * 4;
    ok = ok && hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_asc), bytes, hipHostMallocMapped), "hipHostMalloc asc");
    if (ok) memset(e->h_asc, 0, bytes);
    ok = ok && hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&e->d_asc), e->h_asc, 0), "map asc");
  }
  // (the 2 .. 32-state family: where it runs operation chains -- four rate categories, per-site scaling)
  const bool class_family = e->family == KernelFamily::S20 || e->family == KernelFamily::S4 ||
                            (e->family == KernelFamily::S16 && chains_supported_s16(e));
  // Tips that are vectors (no PLL_ATTRIB_PATTERN_TIP) but came through pll_set_tip_states are known per class of sites
  // -- the state masks -- whether or not the partition asks for site repeats: the operations above them read them as
  // wide tips (4 bytes per site and a row table per branch instead of R * S * 8 bytes per site; bit for bit the same
  // values: upload_tip_classes).  PLLHIP_TIP_CLASSES=0: tips as vectors only.
  static const int env_tip_classes = getenv("PLLHIP_TIP_CLASSES") ? atoi(getenv("PLLHIP_TIP_CLASSES")) : 1;
  e->tip_classes = env_tip_classes != 0 && !e->coded_tips && !e->rate_scalers && class_family;
  if ((e->site_repeats || e->tip_classes) && class_family)
  {
    e->cherries.assign(e->nodes, Engine::Cherry());
    e->tip_version.assign(e->tips, 0);
    e->scaler_lazy.assign(e->nscalers, -1);
  }
  if (!class_family) e->site_repeats = false;   // (the 61-state family, other rate counts of the 2 .. 32-state one: not yet)
  e->pmat_brlen.assign(e->nmat, std::numeric_limits<double>::quiet_NaN());
  e->pmat_params.assign(e->nmat, std::vector<unsigned>());
  e->owner = p;
  e->transient_live.assign(e->nodes, 0);
  e->transient_op.assign(e->nodes, pll_operation_t());
  e->transient_mat_users.assign(e->nmat, 0);
  e->transient_mode = getenv("PLLHIP_TRANSIENT") && atoi(getenv("PLLHIP_TRANSIENT")) != 0;

  // model block layout
  size_t off = 0;
  e->off_rates = off;   off += e->R;
  e->off_weights = off; off += e->R;
  e->off_pinv = off;    off += e->nrm;
  e->off_freqs = off;   off += (size_t)e->nrm * e->Sp;
  e->off_evals = off;   off += (size_t)e->nrm * e->Sp;
  e->off_evecs = off;   off += (size_t)e->nrm * e->S * e->Sp;
  e->off_ievecs = off;  off += (size_t)e->nrm * e->S * e->Sp;
  e->model_len = off;
  ok = ok && dev_alloc(&e->d_model, e->model_len, "model block");
  e->model_shadow.assign(e->model_len, std::numeric_limits<double>::quiet_NaN());

  if (!ok)
  {
    engine_destroy(e);
    return nullptr;
  }
  p->engine = e;
  if (!upload_weights(p))
  {
    p->engine = nullptr;
    engine_destroy(e);
    return nullptr;
  }
  return e;
}

void engine_destroy(Engine * e)
{
  if (!e) return;
  if (!e->shards.empty())
  {
    for (pll_partition_t * c : e->shards) pll_partition_destroy(c);
    delete e;
    return;
  }
  (void)hipSetDevice(e->device);
  if (e->counted) --engines_on_device[e->device & 63];
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (double * ptr : e->d_clv) if (ptr) (void)hipFree(ptr);
  for (uint8_t * ptr : e->d_codes) if (ptr) (void)hipFree(ptr);
  for (auto & kv : e->sumtables) (void)hipFree(kv.second);
  (void)hipFree(e->d_scalers);
  (void)hipFree(e->d_pmat);
  (void)hipFree(e->d_pfrag);
  (void)hipFree(e->d_lut);
  (void)hipFree(e->d_s61_votes);
  (void)hipFree(e->d_s61_ttscale);
  for (auto & slot : e->s61_pred) { if (slot.buf[0]) (void)hipFree(slot.buf[0]); if (slot.buf[1]) (void)hipFree(slot.buf[1]); }
  (void)hipFree(e->d_weights);
  (void)hipFree(e->d_invariant);
  (void)hipFree(e->d_tipmap);
  (void)hipFree(e->d_model);
  (void)hipFree(e->d_partials);
  (void)hipFree(e->d_persite);
  for (auto & c : e->cherries) { (void)hipFree(c.table); (void)hipFree(c.pair); (void)hipFree(c.flags); (void)hipFree(c.rep); (void)hipFree(c.counts); }
  (void)hipFree(e->d_class_seen);
  if (e->h_class_total) (void)hipHostFree(e->h_class_total);
  (void)hipFree(e->d_pairlut);
  (void)hipFree(e->d_newton);
  if (e->newton_ready) (void)hipEventDestroy(e->newton_ready);
  if (e->newton_done) (void)hipEventDestroy(e->newton_done);
  if (e->h_newton) (void)hipHostFree(e->h_newton);
  (void)hipFree(e->d_sum_scratch);
  if (e->h_result) (void)hipHostFree(e->h_result);
  if (e->h_asc) (void)hipHostFree(e->h_asc);
  (void)hipFree(e->d_counter);
  batch_free_hook(e->batch);
  (void)hipFree(e->plan.d_buf);
  if (e->plan.h_stage) (void)hipHostFree(e->plan.h_stage);
  if (e->plan.copied) (void)hipEventDestroy(e->plan.copied);
  for (auto & ev : e->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// the small model arrays of a sharded partition live in the parent (the caller writes them
// there); every routed call that depends on them copies them to the shards first
static void push_model(const pll_partition_t * p, pll_partition_t * c)
{
  const size_t S = p->states, Sp = p->states_padded;
  memcpy(c->rates, p->rates, sizeof(double) * p->rate_cats);
  memcpy(c->rate_weights, p->rate_weights, sizeof(double) * p->rate_cats);
  memcpy(c->prop_invar, p->prop_invar, sizeof(double) * p->rate_matrices);
  memcpy(c->eigen_decomp_valid, p->eigen_decomp_valid, sizeof(int) * p->rate_matrices);
  for (unsigned m = 0; m < p->rate_matrices; ++m)
  {
    memcpy(c->frequencies[m], p->frequencies[m], sizeof(double) * Sp);
    memcpy(c->subst_params[m], p->subst_params[m], sizeof(double) * S * (S - 1) / 2);
    // an eigen-system that differs from the shard's copy (the parent decomposed again, or a caller wrote
    // the arrays): the shard's next model check has to look at it, also inside a burst of P-matrix requests
    if (memcmp(c->eigenvals[m], p->eigenvals[m], sizeof(double) * Sp) != 0 ||
        memcmp(c->eigenvecs[m], p->eigenvecs[m], sizeof(double) * S * Sp) != 0 ||
        memcmp(c->inv_eigenvecs[m], p->inv_eigenvecs[m], sizeof(double) * S * Sp) != 0)
    {
      memcpy(c->eigenvals[m], p->eigenvals[m], sizeof(double) * Sp);
      memcpy(c->eigenvecs[m], p->eigenvecs[m], sizeof(double) * S * Sp);
      memcpy(c->inv_eigenvecs[m], p->inv_eigenvecs[m], sizeof(double) * S * Sp);
      engine_of(c)->eigen_touched = true;
    }
  }
}

int upload_weights(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (!e) return PLL_SUCCESS;   // called from the constructor before the engine exists
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
      pll_set_pattern_weights(e->shards[k], p->pattern_weights + e->shard_first[k]);
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  // the ascertainment-bias columns weigh nothing in the sums over the alignment: their part of
  // the likelihood is the correction the host applies (the host array keeps their state weights)
  e->weights_shadow.assign(p->pattern_weights, p->pattern_weights + e->N);
  for (unsigned n = e->Nreal; n < e->N; ++n) e->weights_shadow[n] = 0;
  if (e->N)
  {
    PLLHIP_TRY(hipMemcpyAsync(e->d_weights, e->weights_shadow.data(), (size_t)e->N * sizeof(unsigned),
                              hipMemcpyHostToDevice, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));      // the source is this engine's own buffer: keep it simple
  }
  return PLL_SUCCESS;
}

static int upload_tipmap(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (!lut_active(e) || e->tipmap_codes_uploaded == codes_in_use(e, p)) return PLL_SUCCESS;
  if (e->shadow_codes)
  {
    std::vector<unsigned long long> tm(PLL_ASCII_SIZE, 0ULL);
    std::copy(e->shadow_tipmap.begin(), e->shadow_tipmap.end(), tm.begin());
    PLLHIP_TRY(hipMemcpyAsync(e->d_tipmap, tm.data(), PLL_ASCII_SIZE * sizeof(unsigned long long), hipMemcpyHostToDevice, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));               // (a local source)
  }
  else
    PLLHIP_TRY(hipMemcpyAsync(e->d_tipmap, p->tipmap, PLL_ASCII_SIZE * sizeof(unsigned long long),
                              hipMemcpyHostToDevice, e->stream));
  e->tipmap_codes_uploaded = codes_in_use(e, p);
  return PLL_SUCCESS;
}

int upload_tip_codes(pll_partition_t * p, unsigned tip)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    // the shards take the parent's codes and code table as they are (a checkpoint loader wrote them)
    for (size_t k = 0; k < e->shards.size(); ++k)
    {
      pll_partition_t * c = e->shards[k];
      memcpy(c->tipchars[tip], p->tipchars[tip] + e->shard_first[k], c->sites);
      memcpy(c->tipmap, p->tipmap, sizeof(pll_state_t) * PLL_ASCII_SIZE);
      memcpy(c->charmap, p->charmap, PLL_ASCII_SIZE);
      if (c->maxstates != p->maxstates) { c->maxstates = p->maxstates; invalidate_luts(c); }
      engine_of(c)->tipmap_codes_uploaded = ~0u;
      if (!upload_tip_codes(c, tip)) return PLL_FAILURE;
    }
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!transient_flush_all(e)) return PLL_FAILURE;           // vectors that exist as their operation only read the old codes
  if (e->N)
    PLLHIP_TRY(hipMemcpyAsync(e->d_codes[tip], p->tipchars[tip], (size_t)e->N,
                              hipMemcpyHostToDevice, e->stream));
  if (tip < e->tip_version.size()) e->tip_version[tip] = ++e->class_clock;     // class maps above this tip are stale
  return upload_tipmap(p);
}

// device staging buffer that frees itself on every exit path
struct DevTmp
{
  double * ptr = nullptr;
  ~DevTmp() { if (ptr) (void)hipFree(ptr); }
};

// API layout [site][rate][Sp] on the host -> device layout of the family
static int store_clv(Engine * e, double * d_dst, const double * host_clv)
{
  const size_t len = (size_t)e->N * e->R * e->Sp;
  if (!len) return PLL_SUCCESS;
  if (!e->blocked)
  {
    PLLHIP_TRY(hipMemcpyAsync(d_dst, host_clv, len * sizeof(double), hipMemcpyHostToDevice, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    return PLL_SUCCESS;
  }
  DevTmp tmp;
  if (!dev_alloc(&tmp.ptr, len, "layout staging")) return PLL_FAILURE;
  PLLHIP_TRY(hipMemcpyAsync(tmp.ptr, host_clv, len * sizeof(double), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_s20_to_blocked, dim3(e->cu_count * 8), dim3(256), 0, e->stream,
                     tmp.ptr, d_dst, e->N, e->nblk, e->R, e->Sp, e->rows);
  PLLHIP_TRY(hipGetLastError());
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  return PLL_SUCCESS;
}

// device layout of the family -> API layout on the host
static int fetch_clv(Engine * e, const double * d_src, double * host_out)
{
  const size_t len = (size_t)e->N * e->R * e->Sp;
  if (!len) return PLL_SUCCESS;
  if (!e->blocked)
  {
    PLLHIP_TRY(hipMemcpyAsync(host_out, d_src, len * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    return PLL_SUCCESS;
  }
  DevTmp tmp;
  if (!dev_alloc(&tmp.ptr, len, "layout staging")) return PLL_FAILURE;
  hipLaunchKernelGGL(k_s20_from_blocked, dim3(e->cu_count * 8), dim3(256), 0, e->stream,
                     d_src, tmp.ptr, e->N, e->R, e->Sp, e->rows);
  PLLHIP_TRY(hipGetLastError());
  PLLHIP_TRY(hipMemcpyAsync(host_out, tmp.ptr, len * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  return PLL_SUCCESS;
}

int upload_tip_clv(pll_partition_t * p, unsigned tip, const double * host_clv)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!upload_tip_clv(e->shards[k], tip, host_clv + (size_t)e->shard_first[k] * e->R * e->Sp)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!transient_flush_all(e)) return PLL_FAILURE;
  if (e->shadow_codes) e->tip_has_codes[tip] = 0;               // (pll_set_tip_states brings the codes afterwards)
  if (!e->cherries.empty())
  {
    // whatever classes the tip had are not those of the new vector (pll_set_tip_states tells them afterwards)
    e->cherries[tip].valid = false;
    e->tip_version[tip] = ++e->class_clock;
    e->plan.key.clear();
  }
  return store_clv(e, e->d_clv[tip], host_clv);
}

// Site repeats without PLL_ATTRIB_PATTERN_TIP (the combination libpll itself allows; the reference's harness:
// test/src/common.c:15-31, "tv" and "sr" are separate switches): a tip set through pll_set_tip_states is a vector
// AND a class node -- the class of a site is its state mask, the table the masks' 0/1 vectors -- so that the
// operations above it are computed per class exactly as above coded tips: class maps from its classes, its rows
// through the P-matrix of its branch with the MFMA sequence of an inner child (bit for bit what the vector gives).
// site_class[n] < nclasses, masks[class] = the state set.
int upload_tip_classes(pll_partition_t * p, unsigned tip, const unsigned * site_class, const unsigned long long * masks,
                       unsigned nclasses)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!upload_tip_classes(e->shards[k], tip, site_class + e->shard_first[k], masks, nclasses)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  if (e->shadow_codes && nclasses && e->N)
  {
    // 33 .. 64 states: the tip's classes become codes of the engine's own table (the state masks met so far, all tips)
    PLLHIP_TRY(hipSetDevice(e->device));
    std::vector<unsigned> code_of(nclasses, 0u);
    const size_t before = e->shadow_tipmap.size();
    for (unsigned k = 0; k < nclasses; ++k)
    {
      size_t c = 0;
      for (; c < e->shadow_tipmap.size(); ++c) if (e->shadow_tipmap[c] == masks[k]) break;
      if (c == e->shadow_tipmap.size())
      {
        if (c >= 255) { e->shadow_tipmap.resize(before); return PLL_SUCCESS; }    // (too many codes: the tip stays a vector)
        e->shadow_tipmap.push_back(masks[k]);
      }
      code_of[k] = (unsigned)c;
    }
    std::vector<uint8_t> codes(e->Nalloc, 0);
    for (unsigned n = 0; n < e->N; ++n) codes[n] = (uint8_t)code_of[site_class[n] < nclasses ? site_class[n] : 0u];
    PLLHIP_TRY(hipMemcpyAsync(e->d_codes[tip], codes.data(), codes.size(), hipMemcpyHostToDevice, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    e->tip_has_codes[tip] = 1;
    if (e->shadow_tipmap.size() != before) { e->lut_stale = true; e->plan.key.clear(); }
    return upload_tipmap(p);
  }
  if (e->coded_tips || e->cherries.empty() || !nclasses || !e->N) return PLL_SUCCESS;
  PLLHIP_TRY(hipSetDevice(e->device));
  Engine::Cherry & c = e->cherries[tip];
  if (!cherry_storage(e, tip, nclasses)) return PLL_FAILURE;
  if (!c.pair && !dev_alloc(&c.pair, (size_t)e->Nalloc, "class codes")) return PLL_FAILURE;
  std::vector<unsigned> cls(e->Nalloc, 0u);                       // (padding sites: class 0)
  for (unsigned n = 0; n < e->N; ++n) cls[n] = site_class[n] < nclasses ? site_class[n] : 0u;
  PLLHIP_TRY(hipMemcpyAsync(c.pair, cls.data(), cls.size() * sizeof(unsigned), hipMemcpyHostToDevice, e->stream));
  // the table: a pseudo alignment of the classes in the API layout, brought into the layout of the family's tables
  const unsigned npblk = (nclasses + S20_BS - 1) / S20_BS;
  std::vector<double> api((size_t)nclasses * e->R * e->Sp, 0.0);
  for (unsigned k = 0; k < nclasses; ++k)
    for (unsigned r = 0; r < e->R; ++r)
      for (unsigned j = 0; j < e->S; ++j)
        api[((size_t)k * e->R + r) * e->Sp + j] = (double)((masks[k] >> j) & 1ULL);
  DevTmp tmp;
  if (e->blocked)
  {
    if (!dev_alloc(&tmp.ptr, api.size(), "class table staging")) return PLL_FAILURE;
    PLLHIP_TRY(hipMemcpyAsync(tmp.ptr, api.data(), api.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_s20_to_blocked, dim3(std::max(1u, std::min(npblk * e->R, e->cu_count * 8u))), dim3(256), 0, e->stream,
                       tmp.ptr, c.table, nclasses, npblk, e->R, e->Sp, e->rows);
    PLLHIP_TRY(hipGetLastError());
  }
  else
    PLLHIP_TRY(hipMemcpyAsync(c.table, api.data(), api.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));                    // (the sources are local)
  (void)hipFree(c.rep); c.rep = nullptr; c.rep_cap = 0;
  c.nclasses = nclasses;
  c.ncodes = 0;
  c.valid = c.materialized = c.map_valid = c.trackable = true;
  c.scaler_index = PLL_SCALE_BUFFER_NONE;
  c.child[0] = c.child[1] = ~0u;
  c.version = e->tip_version[tip] = ++e->class_clock;
  e->plan.key.clear();
  return PLL_SUCCESS;
}

void invalidate_luts(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (e) e->lut_stale = true;
  if (e) for (pll_partition_t * c : e->shards) invalidate_luts(c);
}

// host model arrays -> device if anything changed since the last call.  The arrays are compared
// in place against the shadow of what the device holds (no copy, no allocation: the reference
// issues one pll_update_prob_matrices call per branch, 397 per C3 evaluation).
// light: only the arrays a caller pokes between two consecutive P-matrix requests (rates, weights,
// p-inv, frequencies); the eigen-systems -- 2 S Sp doubles per rate matrix, 63 KiB at 61 states --
// were compared by the first request of the burst and change only through ensure_eigen, which
// runs before this.
int sync_model(pll_partition_t * p, bool light)
{
  Engine * e = engine_of(p);
  const double * sh = e->model_shadow.data();
  bool same = e->model_shadow.size() == e->model_len &&
              memcmp(sh + e->off_rates, p->rates, sizeof(double) * e->R) == 0 &&
              memcmp(sh + e->off_weights, p->rate_weights, sizeof(double) * e->R) == 0 &&
              memcmp(sh + e->off_pinv, p->prop_invar, sizeof(double) * e->nrm) == 0;
  for (unsigned m = 0; same && m < e->nrm; ++m)
  {
    same = memcmp(sh + e->off_freqs + (size_t)m * e->Sp, p->frequencies[m], sizeof(double) * e->Sp) == 0;
    if (same && !(light && !e->eigen_touched))
      same = memcmp(sh + e->off_evals + (size_t)m * e->Sp, p->eigenvals[m], sizeof(double) * e->Sp) == 0 &&
             memcmp(sh + e->off_evecs + (size_t)m * e->S * e->Sp, p->inv_eigenvecs[m], sizeof(double) * e->S * e->Sp) == 0 &&
             memcmp(sh + e->off_ievecs + (size_t)m * e->S * e->Sp, p->eigenvecs[m], sizeof(double) * e->S * e->Sp) == 0;
  }
  if (!light) e->pmatrix_burst = false;
  e->eigen_touched = false;
  if (same) return PLL_SUCCESS;

  std::vector<double> cur(e->model_len, 0.0);
  memcpy(&cur[e->off_rates], p->rates, sizeof(double) * e->R);
  memcpy(&cur[e->off_weights], p->rate_weights, sizeof(double) * e->R);
  memcpy(&cur[e->off_pinv], p->prop_invar, sizeof(double) * e->nrm);
  for (unsigned m = 0; m < e->nrm; ++m)
  {
    memcpy(&cur[e->off_freqs + (size_t)m * e->Sp], p->frequencies[m], sizeof(double) * e->Sp);
    memcpy(&cur[e->off_evals + (size_t)m * e->Sp], p->eigenvals[m], sizeof(double) * e->Sp);
    // device slots: evecs = V, ievecs = V^-1; the partition fields follow libpll-2's naming,
    // where `inv_eigenvecs` is V and `eigenvecs` is V^-1 (pll_model.cpp)
    memcpy(&cur[e->off_evecs + (size_t)m * e->S * e->Sp], p->inv_eigenvecs[m], sizeof(double) * e->S * e->Sp);
    memcpy(&cur[e->off_ievecs + (size_t)m * e->S * e->Sp], p->eigenvecs[m], sizeof(double) * e->S * e->Sp);
  }
  // queued P-matrix requests belong to the model state they were issued under,
  // which is the one still on the device
  if (!flush_pmatrices(p)) return PLL_FAILURE;
  PLLHIP_TRY(hipSetDevice(e->device));
  // pageable source: the runtime stages it before returning, so `cur` may die
  PLLHIP_TRY(hipMemcpyAsync(e->d_model, cur.data(), sizeof(double) * e->model_len,
                            hipMemcpyHostToDevice, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  e->model_shadow.swap(cur);
  e->counters.model_uploads++;
  e->model_generation++;
  // pattern weights (4*N bytes) are NOT re-compared here: pll-modules changes them
  // only through pll_set_pattern_weights (SURVEY.md section 0.3 lists the fields it
  // pokes directly; weights are not among them), which uploads them itself.
  return PLL_SUCCESS;
}

// Launch the queued P-matrix requests.  pll-modules asks for one matrix per call
// (src/tree/treeinfo.c:845-865: 2n-3 calls per evaluation); the requests are
// queued on the host and go to the device up to 200 per launch as soon as a
// consumer of P-matrices (partials, lnL, host mirror) or a model change needs them.
int flush_pmatrices(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (e->pend_midx.empty()) return PLL_SUCCESS;
  PLLHIP_TRY(hipSetDevice(e->device));
  // a vector that an evaluate-only traversal did not store is recomputed with the matrices it was made with: stored
  // now, before one of them changes (an operation list gives the vectors it overwrites up BEFORE it comes here)
  if (e->ntransient && !e->transient_busy)
    for (unsigned m : e->pend_midx)
      if (e->transient_mat_users[m])
      {
        if (!transient_flush_all(e)) return PLL_FAILURE;
        break;
      }
  if (lut_active(e))
  {
    // LUT storage must exist so that the kernel can fill it in the same pass
    const bool stale = e->lut_stale || !e->d_lut || e->lut_codes < std::max(1u, codes_in_use(e, p));
    if (stale && !ensure_luts(p)) return PLL_FAILURE;
  }
  const ModelView mv = model_view(e);
  // operands staged in LDS when two S x Sp matrices fit the kernel's register tiling (S <= 64)
  const int staged = (size_t)e->S * e->Sp <= 4096;
  const size_t lds = staged ? sizeof(double) * 2 * (size_t)(e->Sp == 64 ? 64 : e->S) * e->Sp   // 64 columns: whole 64 x 64 operands (matrix cores)
                            : sizeof(double) * ((size_t)e->Sp + (size_t)e->S * e->Sp);
  const unsigned count = (unsigned)e->pend_midx.size();
  for (unsigned base = 0; base < count; base += MAX_PMAT_PER_LAUNCH)
  {
    const unsigned nb = std::min(MAX_PMAT_PER_LAUNCH, count - base);
    PmatBatch batch;
    for (unsigned q = 0; q < nb; ++q)
    {
      batch.midx[q] = e->pend_midx[base + q];
      batch.t[q] = e->pend_t[base + q];
      e->pend_pos[batch.midx[q]] = -1;
    }
    hipLaunchKernelGGL(k_pmatrix, dim3(nb, e->R), dim3(256), lds, e->stream,
                       mv, e->pend_params, batch, e->R, e->d_pmat,
                       lut_active(e) ? e->d_lut : nullptr, e->lut_codes, e->d_tipmap, staged, e->d_pfrag);
    PLLHIP_TRY(hipGetLastError());
    e->counters.pmatrix_launches++;
  }
  e->pend_midx.clear();
  e->pend_t.clear();
  e->pmat_host_dirty = true;
  return PLL_SUCCESS;
}

static int ensure_eigen(pll_partition_t * p, const unsigned * params_indices)
{
  for (unsigned r = 0; r < p->rate_cats; ++r)
  {
    const unsigned idx = params_indices[r];
    if (idx >= p->rate_matrices)
    {
      set_error(PLL_ERROR_PARAM_INVALID, "params index %u out of range", idx);
      return PLL_FAILURE;
    }
    if (!p->eigen_decomp_valid[idx])
    {
      if (!update_eigen_host(p, idx)) return PLL_FAILURE;
      engine_of(p)->eigen_touched = true;           // the next model check looks at the eigen-systems as well
    }
  }
  return PLL_SUCCESS;
}

static ParamIdx make_params(const pll_partition_t * p, const unsigned * idx)
{
  ParamIdx pi;
  memset(&pi, 0, sizeof(pi));
  for (unsigned r = 0; r < p->rate_cats; ++r) pi.v[r] = idx[r];
  return pi;
}

// LUT storage exists once tips are coded; (re)build when the code table grew
static int ensure_luts(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (!lut_active(e)) return PLL_SUCCESS;
  if (!upload_tipmap(p)) return PLL_FAILURE;
  const unsigned want = std::max(1u, codes_in_use(e, p));
  if (!e->d_lut || e->lut_codes < want)
  {
    // grow with head-room so that a few late codes do not re-allocate
    // (but never past the size the 20-/61-state kernels can stage in LDS: 30 / 67 codes)
    unsigned cap = std::min<unsigned>(PLL_ASCII_SIZE, std::max(want, (e->S == 4) ? 16u : want + 8u));
    if (e->family == KernelFamily::S61 && want <= S61_FRAGS / e->S) cap = std::min(cap, S61_FRAGS / e->S);   // 67 at 61 states
    if (e->family == KernelFamily::S20 && want <= 30u) cap = std::min(cap, 30u);
    // (2 .. 32 states: ... nor past what the family stages in LDS, S16_LUT_LDS doubles per table, while the codes in use fit)
    if (e->family == KernelFamily::S16 && e->R * want * e->S <= S16_LUT_LDS) cap = std::min(cap, std::max(want, S16_LUT_LDS / (e->R * e->S)));
    if (e->d_lut) { PLLHIP_TRY(hipStreamSynchronize(e->stream)); (void)hipFree(e->d_lut); e->d_lut = nullptr; }
    e->plan.key.clear();                            // cached schedules point into the old tables
    if (!dev_alloc(&e->d_lut, (size_t)e->nmat * e->R * cap * e->S, "tip lookup tables")) return PLL_FAILURE;
    e->lut_codes = cap;
    e->lut_stale = true;
  }
  if (e->lut_stale && e->nmat)
  {
    hipLaunchKernelGGL(k_rebuild_lut, dim3(e->nmat, e->R), dim3(256), 0, e->stream,
                       e->S, e->Sp, e->R, e->d_pmat, e->d_lut, e->lut_codes, e->d_tipmap);
    PLLHIP_TRY(hipGetLastError());
  }
  e->lut_stale = false;
  return PLL_SUCCESS;
}

static int ensure_invariant(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  bool need = false;
  for (unsigned m = 0; m < e->nrm; ++m) if (p->prop_invar[m] > 0.0) need = true;
  if (!need || !p->invariant) return PLL_SUCCESS;
  if (!e->d_invariant)
    if (!dev_alloc(&e->d_invariant, (size_t)e->N, "invariant sites")) return PLL_FAILURE;
  if (!e->invariant_uploaded)
  {
    PLLHIP_TRY(hipMemcpyAsync(e->d_invariant, p->invariant, (size_t)e->N * sizeof(int),
                              hipMemcpyHostToDevice, e->stream));
    e->invariant_uploaded = true;
  }
  return PLL_SUCCESS;
}

static NodeRef node_ref(const Engine * e, unsigned clv_index)
{
  NodeRef n;
  n.clv = e->d_clv[clv_index];
  // (the engine's own codes of a tip without PLL_ATTRIB_PATTERN_TIP serve the operations above it only: a reader of the
  // tip itself takes the vector)
  n.codes = (e->coded_tips && clv_index < e->tips) ? e->d_codes[clv_index] : nullptr;
  return n;
}

static const unsigned * scaler_ptr(const Engine * e, int idx)
{
  return (idx == PLL_SCALE_BUFFER_NONE) ? nullptr : e->d_scalers + (size_t)idx * e->sc_len;
}

static unsigned reduce_grid(const Engine * e)
{
  // generic: one thread per site; S4: one lane per (site, rate), 4 per trip;
  // S20: one wave per 32-site block
  unsigned long long need = ((unsigned long long)e->N + 255ULL) / 256ULL;
  if (e->blocked) need = ((unsigned long long)e->nblk + 3ULL) / 4ULL;
  else if (e->family == KernelFamily::S4) need = ((unsigned long long)e->N * e->R + 1023ULL) / 1024ULL;
  // fewer, longer-lived workgroups than one per trip: their prologue (matrix / fragment loads)
  // and the block reduction are paid less often (measured at 1 M sites: 4-state lnL 82 -> 62 us
  // with 1024 blocks, 20-state lnL 254 -> 245 us with 2048)
  static const unsigned env_cap = getenv("PLLHIP_REDUCE_BLOCKS") ? (unsigned)atoi(getenv("PLLHIP_REDUCE_BLOCKS")) : 0u;
  const unsigned cap = env_cap ? env_cap : (e->family == KernelFamily::S4 ? 1024u : 2048u);
  return (unsigned)std::max<unsigned long long>(1ULL, std::min<unsigned long long>(need, std::min(cap, (unsigned)REDUCE_BLOCKS)));
}

// --- ascertainment-bias correction on the host (same formulas as oracle/orc_kernels.c) ---------
// from the log-likelihoods l[k] of the S constant patterns:
//   Lewis  - W log(1 - sum_k L_k);  Felsenstein  + w log(sum_k L_k);  Stamatakis  + sum_k w_k log(L_k)
static double asc_correction(const pll_partition_t * p, const double * l)
{
  const unsigned S = p->states;
  const unsigned * w = p->pattern_weights + p->sites;
  unsigned wsum = 0;
  double mx = l[0], sum = 0.0, corr = 0.0;
  for (unsigned k = 0; k < S; ++k) { mx = std::max(mx, l[k]); wsum += w[k]; }
  for (unsigned k = 0; k < S; ++k) sum += exp(l[k] - mx);
  switch (p->attributes & PLL_ATTRIB_AB_MASK)
  {
    case PLL_ATTRIB_AB_LEWIS: return -(double)p->pattern_weight_sum * log1p(-exp(mx) * sum);
    case PLL_ATTRIB_AB_FELSENSTEIN: return (double)wsum * (mx + log(sum));
    case PLL_ATTRIB_AB_STAMATAKIS:
      for (unsigned k = 0; k < S; ++k) corr += (double)w[k] * l[k];
      return corr;
    default: return 0.0;
  }
}

// first and second derivative of the correction from {A, B, C, count} of the constant patterns
static void asc_derivatives(const pll_partition_t * p, const double * abc, double * d1, double * d2)
{
  const unsigned S = p->states;
  const unsigned * w = p->pattern_weights + p->sites;
  unsigned mn = ~0u, wsum = 0;
  double a = 0.0, b = 0.0, c = 0.0;
  *d1 = *d2 = 0.0;
  for (unsigned k = 0; k < S; ++k) { mn = std::min(mn, (unsigned)abc[4 * k + 3]); wsum += w[k]; }
  for (unsigned k = 0; k < S; ++k)
  {
    const unsigned d = (unsigned)abc[4 * k + 3] - mn;
    const double f = (d == 0) ? 1.0 : (d <= 3) ? ldexp(1.0, -256 * (int)d) : 0.0;
    a += f * abc[4 * k]; b += f * abc[4 * k + 1]; c += f * abc[4 * k + 2];
  }
  switch (p->attributes & PLL_ATTRIB_AB_MASK)
  {
    case PLL_ATTRIB_AB_LEWIS:
    {
      const double t = (mn == 0) ? 1.0 : (mn <= 3) ? ldexp(1.0, -256 * (int)mn) : 0.0;
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
      for (unsigned k = 0; k < S; ++k)
      {
        const double r1 = abc[4 * k + 1] / abc[4 * k], r2 = abc[4 * k + 2] / abc[4 * k];
        *d1 += (double)w[k] * r1;
        *d2 += (double)w[k] * (r2 - r1 * r1);
      }
      break;
    default: break;
  }
}

// Scalar results.  A reduction launch leaves its totals in the engine's current sink
// (Engine::sink): by default the pinned, device-mapped result buffer, whose sequence word
// the host polls; a deferred result (pllhip_results_*) points the sink at a device slot.
static void sink_to_host(Engine * e)
{
  e->sink.dst = e->d_result;
  e->sink.flag = reinterpret_cast<unsigned long long *>(e->d_result) + RESULT_SEQ_SLOT;
  e->sink.seq = ++e->result_seq;
}

// after the reduction kernel: the two-launch form adds the block totals in a second,
// single-block kernel; the fused form has nothing left to launch
static int finish_launch(Engine * e, unsigned nblocks, unsigned n_quant)
{
  if (e->fused_finish) return PLL_SUCCESS;
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, e->stream, reduce_out(e), nblocks, n_quant);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

// wait until the host sees sequence number `seq` in `flag` (mapped memory): poll for a short
// while -- scalar-returning calls are latency-bound for small slices (Newton-Raphson: a
// 14 us kernel per call) -- then fall back to a blocking stream synchronisation.
// PLLHIP_SPIN_US=0 disables the polling.
int wait_sequence(hipStream_t stream, const volatile unsigned long long * flag, unsigned long long seq, double timeout_s)
{
  static const long spin_us = getenv("PLLHIP_SPIN_US") ? atol(getenv("PLLHIP_SPIN_US")) : 400;
  bool done = false;
  const auto t0 = std::chrono::steady_clock::now();
  if (spin_us > 0)
  {
    for (unsigned it = 0; !done; ++it)
    {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) done = true;
      else if ((it & 63u) == 63u &&
               std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us)
        break;
    }
  }
  if (!done && timeout_s > 0.0)
  {
    // a result another rank contributes to (a collective sits in front of it): never block without a bound --
    // a peer that died leaves the collective's kernel spinning for ever.  Poll the stream and the flag.
    for (;;)
    {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return PLL_SUCCESS;
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) break;                      // the stream has drained: the flag decides below
      if (q != hipErrorNotReady) { (void)hip_ok(q, "hipStreamQuery"); return PLL_FAILURE; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      {
        set_error(PLL_ERROR_HIP_TIMEOUT, "no result after %.1f s: a collective did not complete (peer lost?)", timeout_s);
        return PLL_FAILURE;
      }
      struct timespec ts = {0, 50000};
      nanosleep(&ts, nullptr);
    }
  }
  else if (!done) PLLHIP_TRY(hipStreamSynchronize(stream));
  if (!done && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
  {
    set_error(PLL_ERROR_HIP_RUNTIME, "a reduction finished without publishing its result");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

static int finish_reduction(Engine * e, unsigned nblocks, unsigned n_quant, double * out)
{
  if (!finish_launch(e, nblocks, n_quant)) return PLL_FAILURE;
  const volatile unsigned long long * flag =
      reinterpret_cast<const volatile unsigned long long *>(e->h_result) + RESULT_SEQ_SLOT;
  if (!wait_sequence(e->stream, flag, e->sink.seq)) return PLL_FAILURE;
  for (unsigned q = 0; q < n_quant; ++q) out[q] = e->h_result[q];
  return PLL_SUCCESS;
}

static double * sumtable_device(Engine * e, const void * key, bool create)
{
  for (auto it = e->sumtables.begin(); it != e->sumtables.end(); ++it)
    if (it->first == key)
    {
      e->sumtables.splice(e->sumtables.begin(), e->sumtables, it);   // most recently used first
      return e->sumtables.front().second;
    }
  if (!create) return nullptr;
  double * buf = nullptr;
  if (e->sumtables.size() >= MAX_SUMTABLES)
  {
    // recycle the least recently used table (same size for every key)
    buf = e->sumtables.back().second;
    e->sumtables.pop_back();
  }
  else if (!dev_alloc(&buf, e->clv_len, "sumtable"))
    return nullptr;
  e->sumtables.emplace_front(key, buf);
  return buf;
}

static int check_clv_index(const Engine * e, unsigned idx, const char * what)
{
  if (idx >= e->nodes)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "%s CLV index %u out of range", what, idx);
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

static int check_scaler_index(const Engine * e, int idx)
{
  if (idx != PLL_SCALE_BUFFER_NONE && (idx < 0 || (unsigned)idx >= e->nscalers))
  {
    set_error(PLL_ERROR_PARAM_INVALID, "scaler index %d out of range", idx);
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

// ---------------------------------------------------------------------------
// partials: level scheduling + launch
// ---------------------------------------------------------------------------
static int launch_partials(Engine * e, const OpBatch & batch, unsigned nops)
{
  switch (e->family)
  {
    case KernelFamily::S4:
      return launch_partials_s4(e, batch, nops);
    case KernelFamily::S20:
      return launch_partials_s20(e, batch, nops);
    case KernelFamily::S61:
      return launch_partials_s61(e, batch, nops);
    case KernelFamily::S16:
      return launch_partials_s16(e, batch, nops);
    default:
      break;
  }
  return launch_partials_generic(e, batch, nops);
}

// Chain schedule of an operation list (see ChainBatch, engine.h).  Accepted only for
// lists with the shape of a tree traversal -- every vector written once, read by at
// most one later operation, nothing overwritten after it was read, child scalers
// being the ones the producers wrote -- anything else keeps the level schedule.
struct ChainPlan
{
  std::vector<std::vector<unsigned>> chains;     // op indices, bottom to top
  std::vector<int> launch;                       // per chain: launch round
  std::vector<unsigned> lds;                     // per chain: LDS doubles of its operations' tables
  std::vector<unsigned char> carried;            // per op
  int rounds = 0;
};

// LDS doubles of the rows of a wide tip (a class node read through its row table, kernels_repeats.hpp) with `rows`
// classes: staged when they take no more than a coded tip's table may, else gathered from memory (0)
static unsigned wide_slot(const Engine * e, unsigned rows)
{
  if (e->family == KernelFamily::S20)
    return e->R * rows * S20_LUT_RS <= 2560u ? ((e->R * rows * S20_LUT_RS + 7u) & ~7u) : 0u;
  if (e->family == KernelFamily::S16)
    return e->R * rows * e->S <= S16_LUT_LDS ? ((e->R * rows * e->S + 7u) & ~7u) : 0u;
  return 0u;
}

// LDS doubles the tables of operation `op` take in a chain kernel (20 states; 0 otherwise)
// wide1 / wide2: the child is read as a wide tip
static unsigned chain_op_lds(const Engine * e, const pll_operation_t & op, unsigned lut_used, bool wide1 = false, bool wide2 = false)
{
  const bool t1 = e->coded_tips && op.child1_clv_index < e->tips;
  const bool t2 = e->coded_tips && op.child2_clv_index < e->tips;
  if (e->family != KernelFamily::S16 && e->family != KernelFamily::S20) return 0u;
  const bool s16 = e->family == KernelFamily::S16;
  const unsigned a = wide1 ? wide_slot(e, e->cherries[op.child1_clv_index].nclasses) : s16 ? s16_chain_slot(e, t1) : s20_chain_slot(e, t1, lut_used);
  const unsigned b = wide2 ? wide_slot(e, e->cherries[op.child2_clv_index].nclasses) : s16 ? s16_chain_slot(e, t2) : s20_chain_slot(e, t2, lut_used);
  return a + b;
}

// wide: [2 * count] which children are read as wide tips (null: none)
static bool plan_chains(const Engine * e, const pll_operation_t * ops, unsigned count, unsigned max_len,
                        unsigned lds_cap, unsigned lut_used, ChainPlan & plan, const std::vector<unsigned char> * wide = nullptr)
{
  std::vector<int> producer(e->nodes, -1), sc_writer(e->nscalers, -1), chain_of(count, -1);
  std::vector<char> read_ext(e->nodes, 0), sc_read_ext(e->nscalers, 0);
  std::vector<unsigned> size(count, 1), consumers(count, 0);
  plan.carried.assign(count, 0);
  for (unsigned k = 0; k < count; ++k)
  {
    const pll_operation_t & op = ops[k];
    const unsigned child[2] = {op.child1_clv_index, op.child2_clv_index};
    const int child_sc[2] = {op.child1_scaler_index, op.child2_scaler_index};
    if (child[0] == child[1] || op.parent_clv_index == child[0] || op.parent_clv_index == child[1]) return false;
    int pr[2];
    for (int c = 0; c < 2; ++c)
    {
      pr[c] = producer[child[c]];
      if (pr[c] < 0)
      {
        read_ext[child[c]] = 1;
        if (child_sc[c] >= 0)
        {
          if (sc_writer[child_sc[c]] >= 0) return false;
          sc_read_ext[child_sc[c]] = 1;
        }
      }
      else
      {
        if (++consumers[pr[c]] > 1) return false;
        if (child_sc[c] != ops[pr[c]].parent_scaler_index) return false;
      }
    }
    if (producer[op.parent_clv_index] >= 0 || read_ext[op.parent_clv_index]) return false;
    producer[op.parent_clv_index] = (int)k;
    if (op.parent_scaler_index >= 0)
    {
      if (sc_writer[op.parent_scaler_index] >= 0 || sc_read_ext[op.parent_scaler_index]) return false;
      sc_writer[op.parent_scaler_index] = (int)k;
    }

    size[k] = 1 + (pr[0] >= 0 ? size[pr[0]] : 0) + (pr[1] >= 0 ? size[pr[1]] : 0);
    int heavy = -1;                               // which child (0 / 1) continues a chain
    if (pr[0] >= 0 && (pr[1] < 0 || size[pr[0]] >= size[pr[1]])) heavy = 0;
    else if (pr[1] >= 0) heavy = 1;
    const unsigned cost = chain_op_lds(e, op, lut_used, wide && !wide->empty() && (*wide)[2 * k], wide && !wide->empty() && (*wide)[2 * k + 1]);
    if (heavy >= 0 && (plan.chains[chain_of[pr[heavy]]].size() >= max_len ||
                       plan.lds[chain_of[pr[heavy]]] + cost > lds_cap)) heavy = -1;
    int round = 0;
    for (int c = 0; c < 2; ++c)
      if (pr[c] >= 0 && c != heavy) round = std::max(round, plan.launch[chain_of[pr[c]]] + 1);
    if (heavy >= 0)
    {
      const int ch = chain_of[pr[heavy]];
      plan.chains[ch].push_back(k);
      plan.lds[ch] += cost;
      plan.launch[ch] = std::max(plan.launch[ch], round);
      plan.carried[k] = (unsigned char)(heavy + 1);
      chain_of[k] = ch;
    }
    else
    {
      chain_of[k] = (int)plan.chains.size();
      plan.chains.push_back(std::vector<unsigned>(1, k));
      plan.lds.push_back(cost);
      plan.launch.push_back(round);
    }
    plan.rounds = std::max(plan.rounds, plan.launch[chain_of[k]] + 1);
  }
  return true;
}

// make DevicePlan::bytes resident on the device (stream-ordered; nothing is copied when the
// device already holds exactly these bytes) and point `view` at it
static int upload_plan(DevicePlan & dp, hipStream_t stream, PlanView & view)
{
  const size_t len = dp.bytes.size();
  // a few operations without class jobs: in the kernel arguments (no copy in front of the launch: PlanView, engine.h)
  // (only where the kernel arguments live in device memory -- HIP_FORCE_DEV_KERNARG=1, which the library asks for when it
  // is loaded --: the kernels re-read their entries per site block, and those reads should not cross the bus)
  static const int env_inline = getenv("PLLHIP_PLAN_INLINE") ? atoi(getenv("PLLHIP_PLAN_INLINE"))
                              : (getenv("HIP_FORCE_DEV_KERNARG") && atoi(getenv("HIP_FORCE_DEV_KERNARG")) == 1 ? 1 : 0);
  if (env_inline && len <= PLAN_INLINE_BYTES && dp.nops && !dp.ncherry_jobs && !dp.npair_jobs &&
      len == (size_t)dp.nops * sizeof(PlanOp) + (size_t)dp.nchains * sizeof(PlanChain))
  {
    memset(&view, 0, sizeof(view));
    memcpy(view.inl, dp.bytes.data(), len);
    view.inline_ops = dp.nops * (unsigned)sizeof(PlanOp);
    view.nchains = dp.nchains;
    return PLL_SUCCESS;
  }
  view.inline_ops = 0;
  if (dp.resident != dp.bytes)
  {
    if (len > dp.cap)
    {
      PLLHIP_TRY(hipStreamSynchronize(stream));    // a running traversal may still read the old buffer
      (void)hipFree(dp.d_buf);
      dp.d_buf = nullptr;
      dp.cap = 0;
      const size_t cap = std::max<size_t>(2 * len, 65536);
      if (!dev_alloc(&dp.d_buf, cap, "traversal schedule")) return PLL_FAILURE;
      dp.cap = cap;
    }
    if (!dp.copied) PLLHIP_TRY(hipEventCreateWithFlags(&dp.copied, hipEventDisableTiming));
    else PLLHIP_TRY(hipEventSynchronize(dp.copied));  // the previous upload has left the staging buffer
    if (len > dp.h_cap)
    {
      if (dp.h_stage) (void)hipHostFree(dp.h_stage);
      dp.h_stage = nullptr;
      dp.h_cap = 0;
      const size_t cap = std::max<size_t>(2 * len, 65536);
      PLLHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&dp.h_stage), cap, hipHostMallocDefault));
      dp.h_cap = cap;
    }
    memcpy(dp.h_stage, dp.bytes.data(), len);
    PLLHIP_TRY(hipMemcpyAsync(dp.d_buf, dp.h_stage, len, hipMemcpyHostToDevice, stream));
    PLLHIP_TRY(hipEventRecord(dp.copied, stream));
    dp.resident = dp.bytes;
  }
  view.ops = reinterpret_cast<const PlanOp *>(dp.d_buf);
  view.chains = reinterpret_cast<const PlanChain *>(dp.d_buf + (size_t)dp.nops * sizeof(PlanOp));
  view.nchains = dp.nchains;
  return PLL_SUCCESS;
}

// ---------------------------------------------------------------------------
// site repeats (kernels_repeats.hpp): host side
// ---------------------------------------------------------------------------
// a reader needs the per-site counts of scale buffer `sidx`: written out if they exist per class only
static int need_scaler(Engine * e, int sidx)
{
  if (sidx < 0 || (size_t)sidx >= e->scaler_lazy.size() || e->scaler_lazy[sidx] < 0) return PLL_SUCCESS;
  const Engine::Cherry & c = e->cherries[e->scaler_lazy[sidx]];
  hipLaunchKernelGGL(k_class_scaler_expand, dim3(std::max(1u, std::min((e->Nalloc + 255u) / 256u, 4u * e->cu_count))), dim3(256), 0,
                     e->stream, (const unsigned *)c.counts, (const unsigned *)c.pair, e->Nalloc,
                     e->d_scalers + (size_t)sidx * e->sc_len);
  PLLHIP_TRY(hipGetLastError());
  e->scaler_lazy[sidx] = -1;
  return PLL_SUCCESS;
}

// a reader needs the site-indexed vector of `idx` (and the per-site counts that go with it): expand them if the node
// is kept per class
static int need_clv(Engine * e, unsigned idx)
{
  // (an evaluate-only traversal kept it in registers: recomputed and stored now)
  if (e->ntransient && !e->transient_busy && idx < e->transient_live.size() && e->transient_live[idx] &&
      !transient_materialize(e, std::vector<unsigned>(1, idx))) return PLL_FAILURE;
  if (e->cherries.empty() || idx >= e->cherries.size()) return PLL_SUCCESS;
  Engine::Cherry & c = e->cherries[idx];
  if (c.valid && c.scaler_index >= 0 && (size_t)c.scaler_index < e->scaler_lazy.size() && e->scaler_lazy[c.scaler_index] == (int)idx &&
      !need_scaler(e, c.scaler_index)) return PLL_FAILURE;
  if (!c.valid || c.materialized) return PLL_SUCCESS;
  if (e->family == KernelFamily::S4)
    hipLaunchKernelGGL(k_cherry_expand_s4, dim3(std::max(1u, std::min((e->N * e->R + 255u) / 256u, e->cu_count * 8u))), dim3(256), 0,
                       e->stream, c.table, c.pair, e->N, e->R, e->d_clv[idx]);
  else if (e->family == KernelFamily::S16)
  {
#define PLLHIP_CALL(KK) \
    hipLaunchKernelGGL(k_cherry_expand_s16<KK>, dim3(std::max(1u, std::min((e->nblk + 3) / 4, e->cu_count * 8u))), dim3(256), 0, \
                       e->stream, c.table, c.pair, e->nblk, e->R, e->d_clv[idx])
    PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
  }
  else
    hipLaunchKernelGGL(k_cherry_expand, dim3(std::max(1u, std::min((e->nblk + 3) / 4, e->cu_count * 8u))), dim3(256), 0,
                       e->stream, c.table, c.pair, e->nblk, e->R, e->d_clv[idx]);
  PLLHIP_TRY(hipGetLastError());
  c.materialized = true;
  e->repeat_stats.expansions++;
  return PLL_SUCCESS;
}

// which operations of a list the schedule keeps per class (cherries and the class nodes above them), and which
// children the others read as wide tips
struct RepeatPlan
{
  bool active = false;
  std::vector<pll_operation_t> ops;          // the list without the class operations
  std::vector<unsigned> cherry_ops;          // positions of the class operations in the original list (post-order)
  unsigned ncodes = 0;
};

// limits of the class numbering: pairs of child classes that go through a table of all possible pairs (libpll's
// PLL_REPEATS_LOOKUP_SIZE plays this role in pll_update_repeats; more pairs: a hash table of the pairs that occur, as
// long as its slots fit the prefix sum), and how much smaller than the alignment a class table has to be for the
// node to be worth it (the class pipeline writes and gathers rows per class child: no gain from about a third)
// (measured on one box, ms per traversal at ratio 16 / 8 / 4 / 3: C3 random 22.98 / 23.09 / 22.14 / 22.65, simulated
// 20.92 / 20.68 / 18.75 / 18.71; C2 random 1.976 / 1.973 / 1.990 / 2.115, simulated 1.864 / 1.857 / 2.019 / 2.191:
// the 128-byte vectors of 4 states leave less to win per site than the codes and counts of a class node cost)
constexpr unsigned long long CLASS_MAX_PAIRS = 1024ULL * 4096ULL;
constexpr unsigned CLASS_MIN_RATIO_S20 = 4u, CLASS_MIN_RATIO_S4 = 8u;

static bool cherry_storage(Engine * e, unsigned node, unsigned nclasses)
{
  Engine::Cherry & c = e->cherries[node];
  const unsigned npblk = (nclasses + S20_BS - 1) / S20_BS;
  // (20 states: blocked like a vector over the classes; 4 states: [class][rate][4])
  const size_t table_doubles = e->family == KernelFamily::S4 ? (size_t)nclasses * e->R * 4
                             : e->family == KernelFamily::S16 ? (size_t)npblk * e->R * 4 * s16_ks(e) * S20_BS
                                                               : (size_t)npblk * e->R * S20_UNIT;
  if (c.cap_classes < nclasses)
  {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    (void)hipFree(c.table); (void)hipFree(c.flags); (void)hipFree(c.counts);
    c.table = nullptr; c.flags = nullptr; c.counts = nullptr; c.cap_classes = 0;
    e->plan.key.clear();                          // cached schedules point at the old tables
    if (!dev_alloc(&c.table, table_doubles, "class table") ||
        !dev_alloc(&c.flags, (size_t)npblk * S20_BS, "class flags") ||
        !dev_alloc(&c.counts, (size_t)npblk * S20_BS, "class scaler counts") ||
        !hip_ok(hipMemsetAsync(c.flags, 0, (size_t)npblk * S20_BS, e->stream), "memset flags"))
      return false;
    c.cap_classes = nclasses;
  }
  return true;
}

static unsigned long long class_child_version(const Engine * e, unsigned idx)
{
  return idx < e->tips ? e->tip_version[idx] : e->cherries[idx].version;
}

// a tip that is read through byte codes and the partition's lookup tables (PLL_ATTRIB_PATTERN_TIP); without the
// attribute a tip is a vector -- and, when it came through pll_set_tip_states, a class node as well (upload_tip_classes)
static bool coded_tip(const Engine * e, unsigned idx) { return e->coded_tips && idx < e->tips; }

// the class table of a child was made under the code table in use (a tip's own classes do not depend on it)
static bool class_codes_match(const Engine * e, unsigned idx, unsigned lut_used)
{
  return idx < e->tips || e->cherries[idx].ncodes == lut_used;
}

// classes below a child of a class operation: a coded tip's codes, or the classes of a class node
static unsigned class_child_count(const Engine * e, unsigned idx, unsigned lut_used)
{
  return coded_tip(e, idx) ? lut_used : e->cherries[idx].nclasses;
}

// The class map of `node` as the parent of (c1, c2) -- both coded tips or class nodes with current maps -- :
// kept if it was made from these children, made on the device otherwise (one wait for the class count per new
// map of a node above cherries; a cherry needs none).  True if the node is worth keeping per class.
static bool class_map(Engine * e, unsigned node, unsigned c1, unsigned c2, unsigned lut_used, bool & failed)
{
  Engine::Cherry & c = e->cherries[node];
  failed = false;
  if (c.map_valid && c.ncodes == lut_used && c.child[0] == c1 && c.child[1] == c2 &&
      c.child_version[0] == class_child_version(e, c1) && c.child_version[1] == class_child_version(e, c2))
    return c.trackable;
  const unsigned n1 = class_child_count(e, c1, lut_used), n2 = class_child_count(e, c2, lut_used);
  const unsigned long long pairs = (unsigned long long)n1 * n2;
  c.map_valid = true;
  c.trackable = false;
  c.ncodes = lut_used;
  c.child[0] = c1; c.child[1] = c2;
  c.child_version[0] = class_child_version(e, c1); c.child_version[1] = class_child_version(e, c2);
  c.version = ++e->class_clock;
  e->plan.key.clear();
  if (!pairs) return false;
  failed = true;
  if (!c.pair && !dev_alloc(&c.pair, (size_t)e->Nalloc, "class codes")) return false;
  ClassMapArgs a;
  a.codes1 = coded_tip(e, c1) ? e->d_codes[c1] : nullptr;
  a.codes2 = coded_tip(e, c2) ? e->d_codes[c2] : nullptr;
  a.cls1 = coded_tip(e, c1) ? nullptr : e->cherries[c1].pair;
  a.cls2 = coded_tip(e, c2) ? nullptr : e->cherries[c2].pair;
  a.n1 = n1; a.n2 = n2;
  const dim3 gs(std::max(1u, std::min((e->Nalloc + 255u) / 256u, 8u * e->cu_count)));
  if (coded_tip(e, c1) && coded_tip(e, c2))
  {
    // a cherry: every code pair is a class
    hipLaunchKernelGGL(k_class_cherry, gs, dim3(256), 0, e->stream, a, e->Nalloc, c.pair);
    if (hipGetLastError() != hipSuccess) { set_error(PLL_ERROR_HIP_RUNTIME, "class map launch failed"); return false; }
    c.nclasses = (unsigned)pairs;
    (void)hipFree(c.rep); c.rep = nullptr; c.rep_cap = 0;
    c.trackable = true;
    failed = false;
    return true;
  }
  // the pairs that occur are numbered through a table of all pairs, or through a hash table of 2 N .. 4 N slots
  static const unsigned env_ratio = getenv("PLLHIP_CLASS_MIN_RATIO") ? (unsigned)std::max(1, atoi(getenv("PLLHIP_CLASS_MIN_RATIO"))) : 0u;
  const unsigned ratio = env_ratio ? env_ratio : (e->family == KernelFamily::S4 ? CLASS_MIN_RATIO_S4 : CLASS_MIN_RATIO_S20);
  const unsigned max_classes = (unsigned)std::min<unsigned long long>(pairs, e->N / ratio);
  // (PLLHIP_CLASS_TABLE_PAIRS: the largest table of possible pairs, for tests of the hash numbering on small inputs)
  static const unsigned long long env_pairs = getenv("PLLHIP_CLASS_TABLE_PAIRS") ? strtoull(getenv("PLLHIP_CLASS_TABLE_PAIRS"), nullptr, 10) : 0ULL;
  const bool hashed = pairs > (env_pairs ? std::min(env_pairs, CLASS_MAX_PAIRS) : CLASS_MAX_PAIRS);
  unsigned long long slots = pairs;
  if (hashed)
  {
    slots = 1024;
    while (slots < 2ULL * e->N) slots <<= 1;
    if (slots > CLASS_MAX_PAIRS) { failed = false; return false; }       // (alignments beyond 2 M sites: not numbered)
  }
  if (!max_classes) { failed = false; return false; }
  // scratch: seen[slots] | tile sums [1024] | total | (hashed) keys[slots], 8-byte aligned
  const size_t words = (size_t)slots + 1024 + 4 + (hashed ? 2 * (size_t)slots : 0);
  if (e->class_seen_cap < words)
  {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    (void)hipFree(e->d_class_seen); e->d_class_seen = nullptr; e->class_seen_cap = 0;
    if (!dev_alloc(&e->d_class_seen, words, "class numbering scratch")) return false;
    e->class_seen_cap = words;
  }
  if (!e->h_class_total && !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_class_total), sizeof(unsigned), hipHostMallocDefault), "hipHostMalloc"))
    return false;
  if (c.rep_cap < max_classes)
  {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    (void)hipFree(c.rep); c.rep = nullptr; c.rep_cap = 0;
    if (!dev_alloc(&c.rep, 2 * (size_t)max_classes, "class children")) return false;
    c.rep_cap = max_classes;
  }
  unsigned * seen = e->d_class_seen, * tiles = seen + slots, * total = tiles + 1024;
  unsigned long long * keys = hashed ? reinterpret_cast<unsigned long long *>(seen + ((slots + 1024 + 2 + 1) & ~(size_t)1)) : nullptr;
  const unsigned ntiles = (unsigned)((slots + 4095) / 4096), mask = (unsigned)(slots - 1);
  if (!hip_ok(hipMemsetAsync(seen, 0, (size_t)slots * sizeof(unsigned), e->stream), "memset class scratch")) return false;
  if (hashed)
  {
    if (!hip_ok(hipMemsetAsync(keys, 0, (size_t)slots * sizeof(unsigned long long), e->stream), "memset class scratch")) return false;
    hipLaunchKernelGGL(k_class_hash_insert, gs, dim3(256), 0, e->stream, a, e->N, keys, seen, mask);
  }
  else hipLaunchKernelGGL(k_class_mark, gs, dim3(256), 0, e->stream, a, e->N, seen);
  hipLaunchKernelGGL(k_class_scan_tiles, dim3(ntiles), dim3(1024), 0, e->stream, seen, (unsigned)slots, tiles);
  hipLaunchKernelGGL(k_class_scan_top, dim3(1), dim3(1024), 0, e->stream, tiles, ntiles, total);
  hipLaunchKernelGGL(k_class_scan_apply, dim3(ntiles), dim3(1024), 0, e->stream, seen, (unsigned)slots, tiles, n2, c.rep, c.rep_cap,
                     (const unsigned long long *)keys);
  if (hipGetLastError() != hipSuccess) { set_error(PLL_ERROR_HIP_RUNTIME, "class map launch failed"); return false; }
  if (!hip_ok(hipMemcpyAsync(e->h_class_total, total, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream), "class count") ||
      !hip_ok(hipStreamSynchronize(e->stream), "class count")) return false;
  failed = false;
  c.nclasses = *e->h_class_total;
  if (!c.nclasses || c.nclasses > max_classes) return false;
  if (hashed) hipLaunchKernelGGL(k_class_hash_assign, gs, dim3(256), 0, e->stream, a, e->N, e->Nalloc, (const unsigned long long *)keys,
                                 (const unsigned *)seen, mask, c.pair);
  else hipLaunchKernelGGL(k_class_assign, gs, dim3(256), 0, e->stream, a, e->N, e->Nalloc, (const unsigned *)seen, c.pair);
  if (hipGetLastError() != hipSuccess) { set_error(PLL_ERROR_HIP_RUNTIME, "class map launch failed"); failed = true; return false; }
  c.trackable = true;
  return true;
}

// ---------------------------------------------------------------------------
// evaluate-only traversals (include/pllhip.h, pllhip_set_transient): host side
// ---------------------------------------------------------------------------
// the vector of `node` stops being "its operation only": it is being stored (dead = false) or replaced / given up
static void transient_drop(Engine * e, unsigned node, bool dead)
{
  if (!e->ntransient || !e->transient_live[node]) return;
  const pll_operation_t & o = e->transient_op[node];
  e->transient_live[node] = 0;
  --e->transient_mat_users[o.child1_matrix_index];
  --e->transient_mat_users[o.child2_matrix_index];
  --e->ntransient;
  if (dead) e->transient_stats.discarded++;
}

static void transient_mark(Engine * e, const pll_operation_t & o)
{
  transient_drop(e, o.parent_clv_index, true);
  e->transient_live[o.parent_clv_index] = 1;
  e->transient_op[o.parent_clv_index] = o;
  ++e->transient_mat_users[o.child1_matrix_index];
  ++e->transient_mat_users[o.child2_matrix_index];
  ++e->ntransient;
  e->transient_stats.skipped++;
}

// store the vectors of `want` that exist as their operation only: the operations that made them run again (and,
// below them, the ones of the children that were not stored either), as an ordinary list -- the same kernels on the
// same inputs, so what is stored is what the traversal would have stored
static int transient_materialize(Engine * e, const std::vector<unsigned> & want)
{
  if (!e->ntransient) return PLL_SUCCESS;
  std::vector<pll_operation_t> list;
  std::vector<char> seen(e->nodes, 0);
  std::vector<std::pair<unsigned, unsigned>> stack;   // (node, children looked at)
  for (unsigned w : want)
  {
    if (w >= e->nodes || !e->transient_live[w] || seen[w]) continue;
    seen[w] = 1;
    stack.emplace_back(w, 0u);
    while (!stack.empty())
    {
      const unsigned node = stack.back().first;
      const pll_operation_t & o = e->transient_op[node];
      if (stack.back().second < 2)
      {
        const unsigned c = stack.back().second++ ? o.child2_clv_index : o.child1_clv_index;
        if (e->transient_live[c] && !seen[c]) { seen[c] = 1; stack.emplace_back(c, 0u); }
      }
      else { list.push_back(o); stack.pop_back(); }
    }
  }
  if (list.empty()) return PLL_SUCCESS;
  const bool mode = e->transient_mode;
  e->transient_busy = true;
  e->transient_mode = false;
  const int rc = update_partials_impl(e->owner, list.data(), (unsigned)list.size());
  e->transient_mode = mode;
  e->transient_busy = false;
  if (rc) e->transient_stats.materialized += list.size();
  return rc;
}

static int transient_flush_all(Engine * e)
{
  if (!e->ntransient || e->transient_busy) return PLL_SUCCESS;
  std::vector<unsigned> all;
  for (unsigned n = 0; n < e->nodes; ++n) if (e->transient_live[n]) all.push_back(n);
  return transient_materialize(e, all);
}

// before an operation list runs: vectors it reads that were not stored, and vectors that were not stored whose
// inputs (children, scaler buffers) the list overwrites, are stored first; what the list itself overwrites is given up
static int transient_before_list(Engine * e, const pll_operation_t * ops, unsigned count)
{
  if (!e->ntransient) return PLL_SUCCESS;
  if (!e->transient_busy)
  {
    std::vector<char> written(e->nodes, 0), swritten(e->nscalers, 0);
    std::vector<unsigned> want;
    for (unsigned k = 0; k < count; ++k)
    {
      const unsigned child[2] = {ops[k].child1_clv_index, ops[k].child2_clv_index};
      for (int x = 0; x < 2; ++x)
        if (e->transient_live[child[x]] && !written[child[x]]) want.push_back(child[x]);
      written[ops[k].parent_clv_index] = 1;
      if (ops[k].parent_scaler_index >= 0) swritten[ops[k].parent_scaler_index] = 1;
    }
    for (unsigned t = 0; t < e->nodes; ++t)
    {
      if (!e->transient_live[t] || written[t]) continue;
      const pll_operation_t & o = e->transient_op[t];
      if (written[o.child1_clv_index] || written[o.child2_clv_index] ||
          (o.child1_scaler_index >= 0 && swritten[o.child1_scaler_index]) ||
          (o.child2_scaler_index >= 0 && swritten[o.child2_scaler_index]) ||
          (o.parent_scaler_index >= 0 && swritten[o.parent_scaler_index]))
        want.push_back(t);
    }
    if (!want.empty() && !transient_materialize(e, want)) return PLL_FAILURE;
  }
  for (unsigned k = 0; k < count && e->ntransient; ++k) transient_drop(e, ops[k].parent_clv_index, !e->transient_busy);
  return PLL_SUCCESS;
}

// after the launches of a resident schedule: the operations whose vectors stayed in registers
static void transient_after_list(Engine * e, const pll_operation_t * ops, unsigned count, const DevicePlan & dp)
{
  const PlanOp * pops = reinterpret_cast<const PlanOp *>(dp.bytes.data());
  std::vector<int> by_parent;
  for (unsigned i = 0; i < dp.nops; ++i)
  {
    if (!(pops[i].flags & 1u)) continue;
    if (by_parent.empty())
    {
      by_parent.assign(e->nodes, -1);
      for (unsigned k = 0; k < count; ++k) by_parent[ops[k].parent_clv_index] = (int)k;
    }
    const int k = by_parent[pops[i].d.parent_index];
    if (k >= 0) transient_mark(e, ops[k]);
  }
}

// resolve one operation to device pointers and add its algorithmic bytes
// (child vectors in -- 8*S per (site, rate), a coded tip is 1 byte
// per site --, parent vector out, scalers, the two P-matrices or lookup tables
// once per op) and flops (a 2*S*S matvec per non-tip child + S products)
static void fill_desc(const Engine * e, const pll_operation_t & op, OpDesc & d, double & bytes, double & flops)
{
  const size_t pm_stride = (size_t)e->R * e->S * e->Sp;
  const size_t lut_stride = (size_t)e->R * e->lut_codes * e->S;
  const bool t1 = tip_coded(e, op.child1_clv_index);
  const bool t2 = tip_coded(e, op.child2_clv_index);
  d.clv1 = t1 ? nullptr : e->d_clv[op.child1_clv_index];
  d.codes1 = t1 ? e->d_codes[op.child1_clv_index] : nullptr;
  d.pmat1 = e->d_pmat + pm_stride * op.child1_matrix_index;
  d.pfrag1 = e->d_pfrag ? e->d_pfrag + (size_t)op.child1_matrix_index * e->R * 400 : nullptr;
  d.pfrag2 = e->d_pfrag ? e->d_pfrag + (size_t)op.child2_matrix_index * e->R * 400 : nullptr;
  d.lut1 = t1 ? e->d_lut + lut_stride * op.child1_matrix_index : nullptr;
  d.clv2 = t2 ? nullptr : e->d_clv[op.child2_clv_index];
  d.codes2 = t2 ? e->d_codes[op.child2_clv_index] : nullptr;
  d.pmat2 = e->d_pmat + pm_stride * op.child2_matrix_index;
  d.lut2 = t2 ? e->d_lut + lut_stride * op.child2_matrix_index : nullptr;
  d.scaler1 = scaler_ptr(e, op.child1_scaler_index);
  d.scaler2 = scaler_ptr(e, op.child2_scaler_index);
  d.parent = e->d_clv[op.parent_clv_index];
  d.parent_index = op.parent_clv_index;
  d.child1_index = op.child1_clv_index;
  d.child2_index = op.child2_clv_index;
  d.parent_scaler = const_cast<unsigned *>(scaler_ptr(e, op.parent_scaler_index));
  const double nr = (double)e->N * e->R;
  bytes += nr * 8.0 * e->S * (1.0 + (t1 ? 0.0 : 1.0) + (t2 ? 0.0 : 1.0));
  bytes += (double)e->N * ((t1 ? 1.0 : 0.0) + (t2 ? 1.0 : 0.0));
  bytes += (double)e->N * 4.0 * ((d.scaler1 ? 1 : 0) + (d.scaler2 ? 1 : 0) + (d.parent_scaler ? 1 : 0));
  bytes += 8.0 * e->R * e->S * ((t1 ? (double)e->lut_codes : (double)e->Sp) +
                                (t2 ? (double)e->lut_codes : (double)e->Sp));
  flops += nr * (2.0 * e->S * e->S * ((t1 ? 0.0 : 1.0) + (t2 ? 0.0 : 1.0)) + e->S);
}

// The device-resident schedule of an operation list (chains in dependency order, DevicePlan) in e->plan:
// built unless the cached one was made from the same list and settings.  mode 1: the whole traversal in
// one launch (chains depth first); mode 0: one launch per round of chains.  false: the list does not have
// the shape of a tree traversal (plan_chains).
static std::atomic<unsigned long long> plan_generation{0};

static bool prepare_schedule(Engine * e, const pll_partition_t * p, const pll_operation_t * ops, unsigned count,
                             unsigned mode, const RepeatPlan * rp = nullptr,
                             const pll_operation_t * all_ops = nullptr, unsigned all_count = 0, bool transient = false)
{
  const bool chains20 = e->family == KernelFamily::S20, chains16 = e->family == KernelFamily::S16;
  const bool chains4 = e->family == KernelFamily::S4;
  const unsigned lut_used = std::max(1u, std::min(codes_in_use(e, p), e->lut_codes));
  const unsigned chain_max = chains20 ? S20_CHAIN_MAX : chains16 ? S16_CHAIN_MAX : S4_CHAIN_MAX;
  const unsigned chain_lds = chains20 ? S20_CHAIN_LDS : chains16 ? S16_CHAIN_LDS : ~0u;
  DevicePlan & dp = e->plan;
  const bool by_rounds = mode == 0;
  ChainPlan plan;
  // site repeats: children that are cherries kept per class are read as wide tips (kernels_repeats.hpp)
  std::vector<unsigned char> wide(e->cherries.empty() ? 0 : 2 * (size_t)count, 0);
  unsigned nwide = 0;
  if (!wide.empty())
  {
    std::vector<char> made(e->nodes, 0);
    for (unsigned k = 0; k < count; ++k)
    {
      const unsigned child[2] = {ops[k].child1_clv_index, ops[k].child2_clv_index};
      const int child_scaler[2] = {ops[k].child1_scaler_index, ops[k].child2_scaler_index};
      for (int x = 0; x < 2; ++x)
      {
        // (counts of a buffer that another class node stands for: written out first)
        if (child_scaler[x] >= 0 && e->scaler_lazy[child_scaler[x]] >= 0 && e->scaler_lazy[child_scaler[x]] != (int)child[x] &&
            !need_scaler(e, child_scaler[x])) return false;
        if (coded_tip(e, child[x]) || made[child[x]] || !e->cherries[child[x]].valid) continue;
        // (a cherry built under another code table -- a tip has taken a new ambiguity code since -- is read
        // through its expanded vector: its classes are not the ones this schedule indexes; so is a class node whose
        // counts are asked for under another scale buffer than the one its operation wrote)
        const Engine::Cherry & cc = e->cherries[child[x]];
        if (class_codes_match(e, child[x], lut_used) && (child_scaler[x] == cc.scaler_index)) { wide[2 * k + x] = 1; ++nwide; }
        else if (!need_clv(e, child[x])) return false;
      }
      made[ops[k].parent_clv_index] = 1;
    }
  }
  // (the key holds the list as the caller passed it: the cherries taken out of it are part of the schedule)
  const pll_operation_t * key_ops = all_ops ? all_ops : ops;
  const unsigned key_count = all_ops ? all_count : count;
  // (... and which of its operations are kept per class: the same list can meet other class nodes of earlier calls)
  std::vector<unsigned char> tracked(rp && rp->active ? key_count : 0, 0);
  if (!tracked.empty()) for (unsigned k : rp->cherry_ops) tracked[k] = 1;
  std::vector<unsigned char> key(3 * sizeof(unsigned) + (size_t)key_count * sizeof(pll_operation_t) + wide.size() + tracked.size());
  memcpy(key.data(), &key_count, sizeof(unsigned));
  memcpy(key.data() + sizeof(unsigned), &lut_used, sizeof(unsigned));
  const unsigned mode_key = mode | (transient ? 0x100u : 0u);
  memcpy(key.data() + 2 * sizeof(unsigned), &mode_key, sizeof(unsigned));
  memcpy(key.data() + 3 * sizeof(unsigned), key_ops, (size_t)key_count * sizeof(pll_operation_t));
  if (!wide.empty()) memcpy(key.data() + 3 * sizeof(unsigned) + (size_t)key_count * sizeof(pll_operation_t), wide.data(), wide.size());
  if (!tracked.empty()) memcpy(key.data() + key.size() - tracked.size(), tracked.data(), tracked.size());
  bool have = !dp.key.empty() && dp.key == key;
  const unsigned rep_codes = lut_used;
  if (!have && (nwide || (rp && rp->active)))
  {
    // the row tables of the schedule: one per (class node, branch above it) -- the wide tips of the chains and the
    // class children of the class operations
    size_t rows = 0;
    for (unsigned k = 0; k < count && !wide.empty(); ++k)
    {
      if (wide[2 * k]) rows += e->cherries[ops[k].child1_clv_index].nclasses;
      if (wide[2 * k + 1]) rows += e->cherries[ops[k].child2_clv_index].nclasses;
    }
    if (rp && rp->active)
      for (unsigned k : rp->cherry_ops)
      {
        if (!coded_tip(e, all_ops[k].child1_clv_index)) rows += e->cherries[all_ops[k].child1_clv_index].nclasses;
        if (!coded_tip(e, all_ops[k].child2_clv_index)) rows += e->cherries[all_ops[k].child2_clv_index].nclasses;
      }
    const size_t need = rows * e->R * e->S;
    if (need > e->pairlut_cap)
    {
      if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
      (void)hipFree(e->d_pairlut);
      e->d_pairlut = nullptr;
      e->pairlut_cap = 0;
      if (!dev_alloc(&e->d_pairlut, 2 * need, "wide-tip lookup tables")) return false;
      e->pairlut_cap = 2 * need;
    }
  }
  std::vector<PairLutJob> pair_jobs;             // row tables of the wide tips of the chains
  size_t pairlut_used = 0;
  if (!have && plan_chains(e, ops, count, chain_max, chain_lds, lut_used, plan, &wide))
  {
    // Order of the chains: depth first, so that a vector is consumed soon after it was written
    // (the kernel walks slabs of sites through ALL chains: what a slab wrote a few chains ago
    // is still in L2 / the memory-side cache).  The chains form a tree -- chain c feeds the
    // chain that reads c's last vector as a child from memory -- and the larger feeder goes
    // first, which keeps the number of results waiting for their consumer small.
    const size_t nch = plan.chains.size();
    std::vector<int> chain_at(count, -1), producer_op(e->nodes, -1);
    std::vector<unsigned> weight(nch, 0);
    std::vector<std::vector<size_t>> feeders(nch);
    std::vector<char> is_feeder(nch, 0);
    for (size_t c = 0; c < nch; ++c)
      for (unsigned k : plan.chains[c]) { chain_at[k] = (int)c; producer_op[ops[k].parent_clv_index] = (int)k; }
    for (size_t c = 0; c < nch; ++c)                  // chains are numbered in creation order: feeders first
    {
      weight[c] += (unsigned)plan.chains[c].size();
      for (unsigned k : plan.chains[c])
      {
        const unsigned child[2] = {ops[k].child1_clv_index, ops[k].child2_clv_index};
        for (int x = 0; x < 2; ++x)
        {
          const int pk = producer_op[child[x]];
          if (pk < 0 || pk >= (int)k || chain_at[pk] == (int)c) continue;
          feeders[c].push_back((size_t)chain_at[pk]);
          is_feeder[chain_at[pk]] = 1;
          weight[c] += weight[chain_at[pk]];
        }
      }
    }
    std::vector<size_t> order;
    order.reserve(nch);
    if (by_rounds)
    {
      // round by round, longest chains first within a round (their workgroups are dispatched first)
      for (size_t c = 0; c < nch; ++c) order.push_back(c);
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b)
      {
        if (plan.launch[a] != plan.launch[b]) return plan.launch[a] < plan.launch[b];
        return plan.chains[a].size() > plan.chains[b].size();
      });
    }
    else
    {
      std::vector<std::pair<size_t, size_t>> stack;   // (chain, next feeder)
      for (size_t root = 0; root < nch; ++root)
      {
        if (is_feeder[root]) continue;
        stack.emplace_back(root, 0);
        while (!stack.empty())
        {
          const size_t c = stack.back().first;
          if (stack.back().second == 0)
            std::stable_sort(feeders[c].begin(), feeders[c].end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
          if (stack.back().second < feeders[c].size()) { const size_t f = feeders[c][stack.back().second++]; stack.emplace_back(f, 0); }
          else { order.push_back(c); stack.pop_back(); }
        }
      }
    }
    std::vector<PlanOp> pops(count);
    std::vector<PlanChain> pchains;
    unsigned nops = 0, lds_max = 0, nops_virtual = 0;
    dp.algo_bytes = dp.algo_flops = dp.min_bytes = 0.0;
    dp.launches.clear();
    int cur_round = -1;
    const unsigned extent = chains4 ? e->N : e->nblk;
    const unsigned chain_flags = (chains20 ? s20_chain_lut_lds(e, lut_used) : chains16 ? s16_chain_lut_lds(e) : false) ? 1u : 0u;
    for (size_t c : order)
    {
      const std::vector<unsigned> & ch = plan.chains[c];
      if (dp.launches.empty() || (by_rounds && plan.launch[c] != cur_round))
      {
        if (!dp.launches.empty())
        {
          DevicePlan::Launch & done = dp.launches.back();
          done.end = (unsigned)pchains.size();
          done.ops = nops - done.ops;
          done.bytes = dp.algo_bytes - done.bytes;
          done.flops = dp.algo_flops - done.flops;
          done.min_bytes = dp.min_bytes - done.min_bytes;
        }
        DevicePlan::Launch l;
        l.rows = 1;
        l.begin = (unsigned)pchains.size();
        l.end = l.begin;
        l.ops = nops;                     // running totals until the launch is closed
        l.bytes = dp.algo_bytes;
        l.flops = dp.algo_flops;
        l.min_bytes = dp.min_bytes;
        dp.launches.push_back(l);
        cur_round = plan.launch[c];
      }
      PlanChain pc;
      pc.first = nops;
      pc.len = (unsigned)ch.size();
      pc.extent = extent;
      pc.lut_codes = e->lut_codes;
      pc.lut_used = lut_used;
      pc.flags = chain_flags;
      pchains.push_back(pc);
      unsigned off = 0;
      for (size_t i = 0; i < ch.size(); ++i)
      {
        const pll_operation_t & o = ops[ch[i]];
        PlanOp & po = pops[nops++];
        memset(&po, 0, sizeof(po));
        const double before = dp.algo_bytes;
        fill_desc(e, o, po.d, dp.algo_bytes, dp.algo_flops);
        po.carried = i ? plan.carried[ch[i]] : 0;
        // an evaluate-only traversal: the vectors inside a chain are handed on in registers only
        po.flags = (transient && i + 1 < ch.size()) ? 1u : 0u;
        double wide_saved = (po.flags & 1u) ? (double)e->N * e->R * 8.0 * e->S : 0.0;
        for (int x = 0; x < 2 && !wide.empty(); ++x)
        {
          if (!wide[2 * ch[i] + x]) continue;
          const unsigned cidx = x ? o.child2_clv_index : o.child1_clv_index;
          const unsigned midx = x ? o.child2_matrix_index : o.child1_matrix_index;
          const Engine::Cherry & c = e->cherries[cidx];
          PairLutJob job;
          job.table = c.table;
          job.pfrag = chains4 ? e->d_pmat + (size_t)midx * e->R * 16
                    : chains16 ? e->d_pmat + (size_t)midx * e->R * e->S * e->Sp : e->d_pfrag + (size_t)midx * e->R * 400;
          job.out = e->d_pairlut + pairlut_used;
          job.nrows = c.nclasses;
          pairlut_used += (size_t)e->R * c.nclasses * e->S;
          pair_jobs.push_back(job);
          // wide tip: no vector, no byte codes; pfrag = class codes, lut = its table, childN_index = table rows
          // (... and its scaler counts per class)
          // (bit 1 / 2 of the flags: few rows, staged in LDS by the chain kernels of the 20- and 2 .. 32-state families)
          if (wide_slot(e, c.nclasses)) po.flags |= 2u << x;
          if (x) { po.d.clv2 = nullptr; po.d.codes2 = nullptr; po.d.pfrag2 = reinterpret_cast<const double *>(c.pair); po.d.lut2 = job.out; po.d.child2_index = c.nclasses;
                   if (po.d.scaler2) po.d.scaler2 = c.counts; }
          else   { po.d.clv1 = nullptr; po.d.codes1 = nullptr; po.d.pfrag1 = reinterpret_cast<const double *>(c.pair); po.d.lut1 = job.out; po.d.child1_index = c.nclasses;
                   if (po.d.scaler1) po.d.scaler1 = c.counts; }
          wide_saved += (double)e->N * e->R * 8.0 * e->S - 4.0 * e->N;      // class codes instead of the vector
        }
        // the handed-over child stays in registers: neither its vector nor its scaler counts are read
        dp.min_bytes += dp.algo_bytes - before - wide_saved;
        if (po.carried)
          dp.min_bytes -= (double)e->N * e->R * 8.0 * e->S +
                          ((po.carried == 1 ? po.d.scaler1 : po.d.scaler2) ? 4.0 * (double)e->N * (e->rate_scalers ? e->R : 1) : 0.0);
        const bool t1 = e->coded_tips && o.child1_clv_index < e->tips;
        const bool t2 = e->coded_tips && o.child2_clv_index < e->tips;
        if (chains20 || chains16)
        {
          const bool w1 = !wide.empty() && wide[2 * ch[i]], w2 = !wide.empty() && wide[2 * ch[i] + 1];
          po.slot1 = off;
          off += w1 ? wide_slot(e, e->cherries[o.child1_clv_index].nclasses) : chains20 ? s20_chain_slot(e, t1, lut_used) : s16_chain_slot(e, t1);
          po.slot2 = off;
          off += w2 ? wide_slot(e, e->cherries[o.child2_clv_index].nclasses) : chains20 ? s20_chain_slot(e, t2, lut_used) : s16_chain_slot(e, t2);
        }
      }
      lds_max = std::max(lds_max, chains4 ? (unsigned)ch.size() : off);   // 4 states: the longest chain
    }
    // site repeats: the operations the schedule keeps per class (taken out of the list by the caller), by level
    std::vector<CherryJob> cherry_jobs;
    std::vector<PairLutJob> level_pairs;           // row tables of class children, by level of their parent
    dp.repeat_levels.clear();
    dp.repeat_classes = 0;
    if (rp && rp->active)
    {
      const size_t lut_stride = (size_t)e->R * e->lut_codes * e->S;
      const size_t nrp = rp->cherry_ops.size();
      std::vector<int> node_level(e->nodes, -1);
      std::vector<unsigned> lvl(nrp, 0), order(nrp);
      unsigned max_level = 0;
      for (size_t x = 0; x < nrp; ++x)
      {
        const pll_operation_t & o = all_ops[rp->cherry_ops[x]];
        unsigned l = 0;
        if (o.child1_clv_index >= e->tips) l = std::max(l, 1u + (unsigned)std::max(node_level[o.child1_clv_index], 0));
        if (o.child2_clv_index >= e->tips) l = std::max(l, 1u + (unsigned)std::max(node_level[o.child2_clv_index], 0));
        lvl[x] = l;
        node_level[o.parent_clv_index] = (int)l;
        max_level = std::max(max_level, l);
        order[x] = (unsigned)x;
      }
      std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return lvl[a] < lvl[b]; });
      size_t at = 0;
      for (unsigned l = 0; l <= max_level; ++l)
      {
        DevicePlan::RepeatLevel L = {(unsigned)cherry_jobs.size(), 0, (unsigned)level_pairs.size(), 0, 0, 0};
        for (; at < nrp && lvl[order[at]] == l; ++at)
        {
          const pll_operation_t & o = all_ops[rp->cherry_ops[order[at]]];
          const Engine::Cherry & c = e->cherries[o.parent_clv_index];
          CherryJob j;
          memset(&j, 0, sizeof(j));
          const unsigned child[2] = {o.child1_clv_index, o.child2_clv_index};
          const unsigned matrix[2] = {o.child1_matrix_index, o.child2_matrix_index};
          const int child_scaler[2] = {o.child1_scaler_index, o.child2_scaler_index};
          for (int x = 0; x < 2; ++x)
          {
            const double * rows;
            unsigned nrows;
            if (coded_tip(e, child[x])) { rows = e->d_lut + lut_stride * matrix[x]; nrows = e->lut_codes; }
            else
            {
              const Engine::Cherry & cc = e->cherries[child[x]];
              PairLutJob pj;
              pj.table = cc.table;
              pj.pfrag = chains4 ? e->d_pmat + (size_t)matrix[x] * e->R * 16
                       : chains16 ? e->d_pmat + (size_t)matrix[x] * e->R * e->S * e->Sp : e->d_pfrag + (size_t)matrix[x] * e->R * 400;
              pj.out = e->d_pairlut + pairlut_used;
              pj.nrows = cc.nclasses;
              pairlut_used += (size_t)e->R * cc.nclasses * e->S;
              level_pairs.push_back(pj);
              L.max_rows = std::max(L.max_rows, cc.nclasses);
              rows = pj.out; nrows = cc.nclasses;
            }
            const unsigned * cnt = (child[x] >= e->tips && child_scaler[x] >= 0) ? e->cherries[child[x]].counts : nullptr;
            if (x) { j.lut2 = rows; j.rows2 = nrows; j.cnt2 = cnt; }
            else   { j.lut1 = rows; j.rows1 = nrows; j.cnt1 = cnt; }
          }
          j.rep = c.rep;
          j.nclasses = c.nclasses;
          j.table = c.table; j.flags = c.flags;
          j.counts = o.parent_scaler_index >= 0 ? c.counts : nullptr;
          cherry_jobs.push_back(j);
          L.max_classes = std::max(L.max_classes, c.nclasses);
          dp.repeat_classes += c.nclasses;
          // what SURVEY.md 8d counts for the operation / what it moves now
          OpDesc dummy;
          double ab = 0.0;
          fill_desc(e, o, dummy, ab, dp.algo_flops);
          dp.algo_bytes += ab;
          dp.min_bytes += 2.0 * e->N + 4.0 * e->N;            // (tip codes in once per topology; a class code per site for the consumer)
          ++nops_virtual;
        }
        L.job_end = (unsigned)cherry_jobs.size();
        L.pair_end = (unsigned)level_pairs.size();
        dp.repeat_levels.push_back(L);
      }
    }
    if (!cherry_jobs.empty() || !pair_jobs.empty())
    {
      // ... and the row tables of the wide tips, after the last level
      DevicePlan::RepeatLevel L = {(unsigned)cherry_jobs.size(), (unsigned)cherry_jobs.size(), (unsigned)level_pairs.size(),
                                   (unsigned)(level_pairs.size() + pair_jobs.size()), 0, 0};
      for (const PairLutJob & pj : pair_jobs) L.max_rows = std::max(L.max_rows, pj.nrows);
      dp.repeat_levels.push_back(L);
      level_pairs.insert(level_pairs.end(), pair_jobs.begin(), pair_jobs.end());
    }
    pair_jobs.swap(level_pairs);
    dp.ncherry_jobs = (unsigned)cherry_jobs.size();
    dp.npair_jobs = (unsigned)pair_jobs.size();
    dp.repeat_codes = rep_codes;
    dp.off_cherry_jobs = pops.size() * sizeof(PlanOp) + pchains.size() * sizeof(PlanChain);
    dp.off_pair_jobs = dp.off_cherry_jobs + cherry_jobs.size() * sizeof(CherryJob);
    dp.bytes.resize(dp.off_pair_jobs + pair_jobs.size() * sizeof(PairLutJob));
    memcpy(dp.bytes.data(), pops.data(), pops.size() * sizeof(PlanOp));
    memcpy(dp.bytes.data() + pops.size() * sizeof(PlanOp), pchains.data(), pchains.size() * sizeof(PlanChain));
    if (!cherry_jobs.empty()) memcpy(dp.bytes.data() + dp.off_cherry_jobs, cherry_jobs.data(), cherry_jobs.size() * sizeof(CherryJob));
    if (!pair_jobs.empty()) memcpy(dp.bytes.data() + dp.off_pair_jobs, pair_jobs.data(), pair_jobs.size() * sizeof(PairLutJob));
    if (!dp.launches.empty())
    {
      DevicePlan::Launch & done = dp.launches.back();
      done.end = (unsigned)pchains.size();
      done.ops = nops - done.ops;
      done.bytes = dp.algo_bytes - done.bytes;
      done.flops = dp.algo_flops - done.flops;
      done.min_bytes = dp.min_bytes - done.min_bytes;
    }
    (void)nops_virtual;
    // rounds of ONE chain each that follow one another (the spine towards the root) need no launch
    // boundary between them: one workgroup row walks them in turn, exactly as in a one-launch traversal
    if (by_rounds)
    {
      std::vector<DevicePlan::Launch> merged;
      for (const DevicePlan::Launch & l : dp.launches)
      {
        if (!merged.empty() && l.end - l.begin == 1 && merged.back().rows == 1)
        {
          DevicePlan::Launch & m = merged.back();
          m.end = l.end;
          m.ops += l.ops;
          m.bytes += l.bytes;
          m.flops += l.flops;
          m.min_bytes += l.min_bytes;
        }
        else
        {
          merged.push_back(l);
          merged.back().rows = l.end - l.begin;
        }
      }
      dp.launches.swap(merged);
    }
    else
      for (DevicePlan::Launch & l : dp.launches) l.rows = 1;
    dp.nops = nops;
    dp.nchains = (unsigned)pchains.size();
    dp.lds_doubles = lds_max;
    dp.max_extent = extent;
    dp.generation = ++plan_generation;
    dp.key.swap(key);
    have = true;
  }
  return have;
}

// the class operations and row tables of a resident schedule (kernels_repeats.hpp), on the engine's stream; the
// schedule's job arrays are on the device (upload_plan)
static int launch_class_levels(Engine * e, const DevicePlan & dp)
{
  const bool chains4 = e->family == KernelFamily::S4, chains16 = e->family == KernelFamily::S16;
  // level by level: the row tables of the class children of a level, then the tables of the level (with the
  // scaler counts per class); the last entry holds the row tables of the wide tips of the chains
  const CherryJob * cj = reinterpret_cast<const CherryJob *>(dp.d_buf + dp.off_cherry_jobs);
  const PairLutJob * pj = reinterpret_cast<const PairLutJob *>(dp.d_buf + dp.off_pair_jobs);
  for (size_t lv = 0; lv < dp.repeat_levels.size(); ++lv)
  {
    const DevicePlan::RepeatLevel & L = dp.repeat_levels[lv];
    if (L.pair_end > L.pair_begin)
    {
      const unsigned njobs = L.pair_end - L.pair_begin, npblk = (L.max_rows + S20_BS - 1) / S20_BS;
      const PairLutJob * jobs = pj + L.pair_begin;
      const dim3 gp((npblk + 3) / 4, njobs);
      const size_t lds = sizeof(double) * e->R * S20_CFRAGS;
      if (chains4) hipLaunchKernelGGL(k_pair_lut_s4, dim3((L.max_rows * e->R + 255) / 256, njobs), dim3(256), 0, e->stream, jobs, e->R);
      else if (chains16)
      {
#define PLLHIP_CALL(KK) \
        hipLaunchKernelGGL(k_pair_lut_s16<KK>, gp, dim3(256), sizeof(double) * e->R * s16_fr(KK), e->stream, jobs, e->R, e->S, e->Sp)
        PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
      }
      else if (e->R == 4) hipLaunchKernelGGL(k_pair_lut<4>, gp, dim3(256), lds, e->stream, jobs);
      else if (e->R == 2) hipLaunchKernelGGL(k_pair_lut<2>, gp, dim3(256), lds, e->stream, jobs);
      else hipLaunchKernelGGL(k_pair_lut<1>, gp, dim3(256), lds, e->stream, jobs);
      PLLHIP_TRY(hipGetLastError());
    }
    if (L.job_end > L.job_begin)
    {
      const unsigned njobs = L.job_end - L.job_begin, npblk = (L.max_classes + S20_BS - 1) / S20_BS;
      const CherryJob * jobs = cj + L.job_begin;
      const dim3 gb((npblk + 3) / 4, njobs);
      if (chains4) hipLaunchKernelGGL(k_cherry_build_s4, dim3((L.max_classes + 255) / 256, njobs), dim3(256), 0, e->stream, jobs, e->R, dp.repeat_codes);
      else if (chains16)
      {
#define PLLHIP_CALL(KK) \
        hipLaunchKernelGGL(k_cherry_build_s16<KK>, gb, dim3(256), 0, e->stream, jobs, dp.repeat_codes, e->R, e->S)
        PLLHIP_DISPATCH_KS(s16_ks(e), PLLHIP_CALL);
#undef PLLHIP_CALL
      }
      else if (e->R == 4) hipLaunchKernelGGL(k_cherry_build<4>, gb, dim3(256), 0, e->stream, jobs, dp.repeat_codes);
      else if (e->R == 2) hipLaunchKernelGGL(k_cherry_build<2>, gb, dim3(256), 0, e->stream, jobs, dp.repeat_codes);
      else hipLaunchKernelGGL(k_cherry_build<1>, gb, dim3(256), 0, e->stream, jobs, dp.repeat_codes);
      PLLHIP_TRY(hipGetLastError());
    }
  }
  return PLL_SUCCESS;
}

// index checks of an operation list (the reference interface returns nothing: errors go to pll_errno)
static int validate_ops(const Engine * e, const pll_operation_t * ops, unsigned count)
{
  for (unsigned k = 0; k < count; ++k)
  {
    const pll_operation_t & op = ops[k];
    if (!check_clv_index(e, op.parent_clv_index, "parent") ||
        !check_clv_index(e, op.child1_clv_index, "child1") ||
        !check_clv_index(e, op.child2_clv_index, "child2") ||
        !check_scaler_index(e, op.parent_scaler_index) ||
        !check_scaler_index(e, op.child1_scaler_index) ||
        !check_scaler_index(e, op.child2_scaler_index))
      return PLL_FAILURE;
    if (op.child1_matrix_index >= e->nmat || op.child2_matrix_index >= e->nmat)
    {
      set_error(PLL_ERROR_PARAM_INVALID, "matrix index out of range in operation %u", k);
      return PLL_FAILURE;
    }
    if (op.parent_clv_index < e->tips && e->coded_tips)
    {
      set_error(PLL_ERROR_PARAM_INVALID, "operation %u writes a coded tip", k);
      return PLL_FAILURE;
    }
  }
  return PLL_SUCCESS;
}

static int update_partials_impl(pll_partition_t * p, const pll_operation_t * ops, unsigned count)
{
  Engine * e = engine_of(p);
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!validate_ops(e, ops, count)) return PLL_FAILURE;
  // (before the queued P-matrices are launched: vectors that were not stored and that this list overwrites are given
  // up, the ones it reads are stored with the matrices they were made with; a list that stores such vectors -- 
  // transient_busy -- runs on the matrices the device holds)
  if (!transient_before_list(e, ops, count)) return PLL_FAILURE;
  if (e->shadow_codes)
    for (unsigned k = 0; k < count; ++k)
      if (ops[k].parent_clv_index < e->tips) e->tip_has_codes[ops[k].parent_clv_index] = 0;    // (a tip vector is overwritten)
  if ((!e->transient_busy && !flush_pmatrices(p)) || !ensure_luts(p)) return PLL_FAILURE;

  // dependency levels: an op runs after the producers of its children and
  // after every earlier op that touched its output buffers
  std::vector<int> clv_level(e->nodes, -1), sc_level(e->nscalers, -1), level(count, 0);
  int max_level = 0;
  if (!e->cherries.empty())
  {
    // every vector this list writes stops being the cherry it may have been; one that the list reads first
    // (and that exists per class only) is expanded before it goes
    // ... and the same for scale buffers whose counts exist per class only
    std::vector<char> read(e->nodes, 0), sread(e->nscalers, 0);
    for (unsigned k = 0; k < count; ++k)
    {
      read[ops[k].child1_clv_index] = read[ops[k].child2_clv_index] = 1;
      if (ops[k].child1_scaler_index >= 0) sread[ops[k].child1_scaler_index] = 1;
      if (ops[k].child2_scaler_index >= 0) sread[ops[k].child2_scaler_index] = 1;
      Engine::Cherry & c = e->cherries[ops[k].parent_clv_index];
      if (c.valid && read[ops[k].parent_clv_index] && !need_clv(e, ops[k].parent_clv_index)) return PLL_FAILURE;
      // the node stops being a class node: counts it stands for under a buffer this operation does not rewrite
      // are written out while its table still describes them
      if (c.valid && c.scaler_index >= 0 && e->scaler_lazy[c.scaler_index] == (int)ops[k].parent_clv_index &&
          c.scaler_index != ops[k].parent_scaler_index && !need_scaler(e, c.scaler_index)) return PLL_FAILURE;
      c.valid = false;
      const int sp = ops[k].parent_scaler_index;
      if (sp >= 0 && e->scaler_lazy[sp] >= 0)
      {
        if (sread[sp] && !need_scaler(e, sp)) return PLL_FAILURE;
        e->scaler_lazy[sp] = -1;
      }
    }
  }
  for (unsigned k = 0; k < count; ++k)
  {
    const pll_operation_t & op = ops[k];
    int l = 0;
    l = std::max(l, clv_level[op.child1_clv_index] + 1);
    l = std::max(l, clv_level[op.child2_clv_index] + 1);
    l = std::max(l, clv_level[op.parent_clv_index] + 1);
    if (op.child1_scaler_index >= 0) l = std::max(l, sc_level[op.child1_scaler_index] + 1);
    if (op.child2_scaler_index >= 0) l = std::max(l, sc_level[op.child2_scaler_index] + 1);
    if (op.parent_scaler_index >= 0) l = std::max(l, sc_level[op.parent_scaler_index] + 1);
    // every buffer touched (read or written) is stamped with this level, so any
    // later op that touches it again runs in a later launch.  In a tree
    // traversal a CLV is touched by its producer and its single consumer only,
    // so this costs no parallelism and covers RAW, WAR and WAW alike.
    level[k] = l;
    clv_level[op.parent_clv_index] = l;
    clv_level[op.child1_clv_index] = std::max(clv_level[op.child1_clv_index], l);
    clv_level[op.child2_clv_index] = std::max(clv_level[op.child2_clv_index], l);
    if (op.parent_scaler_index >= 0) sc_level[op.parent_scaler_index] = l;
    if (op.child1_scaler_index >= 0)
      sc_level[op.child1_scaler_index] = std::max(sc_level[op.child1_scaler_index], l);
    if (op.child2_scaler_index >= 0)
      sc_level[op.child2_scaler_index] = std::max(sc_level[op.child2_scaler_index], l);
    max_level = std::max(max_level, l);
  }

  auto prof_begin = [&](hipEvent_t & ev1) -> int
  {
    ev1 = nullptr;
    if (!e->profiling) return PLL_SUCCESS;
    if (e->prof_used == e->prof_events.size())
    {
      hipEvent_t x, y;
      PLLHIP_TRY(hipEventCreate(&x));
      PLLHIP_TRY(hipEventCreate(&y));
      e->prof_events.emplace_back(x, y);
    }
    hipEvent_t ev0 = e->prof_events[e->prof_used].first;
    ev1 = e->prof_events[e->prof_used].second;
    e->prof_used++;
    PLLHIP_TRY(hipEventRecord(ev0, e->stream));
    return PLL_SUCCESS;
  };
  auto prof_end = [&](hipEvent_t ev1, double bytes, double flops, unsigned nops, double min_bytes = -1.0) -> int
  {
    if (!e->profiling) return PLL_SUCCESS;
    PLLHIP_TRY(hipEventRecord(ev1, e->stream));
    e->prof_bytes += bytes;
    e->prof_min_bytes += (min_bytes >= 0.0) ? min_bytes : bytes;
    e->prof_flops += flops;
    e->prof_ops += nops;
    return PLL_SUCCESS;
  };

  // the paths that read child scaler counts per site (everything but the resident schedules, whose wide tips read
  // them per class): counts that exist per class only are written out first
  auto plain_scalers = [&]() -> int
  {
    if (e->scaler_lazy.empty()) return PLL_SUCCESS;
    for (unsigned k = 0; k < count; ++k)
      if (!need_scaler(e, ops[k].child1_scaler_index) || !need_scaler(e, ops[k].child2_scaler_index)) return PLL_FAILURE;
    return PLL_SUCCESS;
  };

  // chain schedule (4- and 20-state families, four rates): one launch per round of chains,
  // the vector of a link stays in registers.  PLLHIP_CHAINS=0 keeps the plain level schedule.
  static const int use_chains = getenv("PLLHIP_CHAINS") ? atoi(getenv("PLLHIP_CHAINS")) : 1;
  const bool chains20 = e->family == KernelFamily::S20 && chains_supported_s20(e);
  const bool chains4 = e->family == KernelFamily::S4 && chains_supported_s4(e);
  // 2..16 states: chains exist in the one-launch form only
  const bool chains16 = e->family == KernelFamily::S16 && chains_supported_s16(e);
  if (use_chains && count >= 2 && (chains20 || chains4 || chains16))
  {
    ChainPlan plan;
    // tip tables are staged with the codes in use (at least one: an untouched partition)
    const unsigned lut_used = std::max(1u, std::min(codes_in_use(e, p), e->lut_codes));
    // PLLHIP_TRAVERSE=1 / 0: whole traversals always / never in one launch (default: by size, below; never
    // when several partitions share the device, each on its own stream: long-lived workgroups with a fixed
    // share of the sites interleave worse than rounds -- two DNA + two protein partitions: 7.7 against 7.1 ms).
    static const int env_traverse = getenv("PLLHIP_TRAVERSE") ? atoi(getenv("PLLHIP_TRAVERSE")) : -1;
    // One launch for the whole traversal, or one launch per round with the chains of the round as grid
    // rows (both from the device-resident schedule)?  In one launch a workgroup walks ALL chains one
    // after the other and no launch boundary is paid; by rounds the chains run side by side (which is
    // what a partition needs that does not fill the chip with its site blocks) and the workgroups of
    // a round are shared out dynamically (which wins again on very large partitions).  Measured per
    // traversal, rounds / one launch:
    //   20 states, 200 taxa:  32 k sites 1.55 / 2.14 ms, 64 k 2.64 / 2.59, 125 k 4.83 / 4.66,
    //                         250 k 8.66 / 8.93, 500 k 16.4 / 17.2, 1 M 32.4 / 33.0
    //    4 states, 100 taxa:  100 k 0.52 / 0.58, 250 k 1.08 / 1.18, 500 k 1.94 / 1.99, 1 M 3.64 / 3.52
    //   16 states,  50 taxa:  8 k 0.23 / 0.38, 32 k 0.36 / 0.44, 128 k 1.03 / 0.94, 500 k 3.22 / 3.17
    //    2 states,  50 taxa:  1 M 2.20 / 2.74
    const bool fills_chip = chains4 ? (e->N + 63) / 64 >= 48u * e->cu_count
                          : chains16 ? (e->nblk >= 12u * e->cu_count && e->nblk < 48u * e->cu_count && e->S > 8)
                                     : (e->nblk >= 6u * e->cu_count && e->nblk < 24u * e->cu_count);
    const bool use_traverse = env_traverse >= 0 ? env_traverse != 0
                            : (engines_on_device[e->device & 63].load() <= 1 && fills_chip && (count >= 6 || chains16));
    const unsigned chain_max = chains20 ? S20_CHAIN_MAX : chains16 ? S16_CHAIN_MAX : S4_CHAIN_MAX;
    const unsigned chain_lds = chains20 ? S20_CHAIN_LDS : chains16 ? S16_CHAIN_LDS : ~0u;
    // Device-resident schedules serve both forms: the whole traversal in one launch, or (lists of six
    // operations and more) one launch per round with the chains of the round as grid rows.  A repeated
    // list is neither planned nor copied again (2.5 us of host time per launch), its kernels read their
    // descriptors from device memory, a round is never split at 24 operations and rounds of single
    // operations stay on the chain kernel: 72 -> 21 us and 212 -> 23 us per full-traversal call of a
    // 2 000-site partition at 4 / 20 states against the by-value form below.
    // Short lists (the 1 - 3 operations of an SPR insertion) keep their descriptors in the kernel
    // arguments: a schedule would have to be copied to the device first (W3 at C2 size: 165 against
    // 178 us per iteration).
    const bool by_rounds = !use_traverse && count >= 6;
    if (use_traverse || by_rounds)
    {
      const unsigned mode = use_traverse ? 1u : 0u;
      DevicePlan & dp = e->plan;
      // site repeats: the cherries of the list that an operation of the list consumes are kept per class
      // (kernels_repeats.hpp) and leave the list; their consumers read them as wide tips
      RepeatPlan rp;
      if (e->site_repeats && (chains20 || chains4 || chains16) && lut_used <= 64)
      {
        // an operation is kept per class if an operation of the list consumes it and both children are known per
        // class: coded tips, class operations earlier in the list, or class nodes of earlier calls whose tables
        // still are their vectors
        std::vector<int> consumer(e->nodes, -1);
        std::vector<char> now(e->nodes, 0);
        for (unsigned k = 0; k < count; ++k) { consumer[ops[k].child1_clv_index] = (int)k; consumer[ops[k].child2_clv_index] = (int)k; }
        for (unsigned k = 0; k < count; ++k)
        {
          const pll_operation_t & o = ops[k];
          const unsigned child[2] = {o.child1_clv_index, o.child2_clv_index};
          const int child_scaler[2] = {o.child1_scaler_index, o.child2_scaler_index};
          bool ok = consumer[o.parent_clv_index] > (int)k && child[0] != child[1];
          for (int x = 0; x < 2 && ok; ++x)
          {
            if (coded_tip(e, child[x])) ok = child_scaler[x] == PLL_SCALE_BUFFER_NONE;
            else
            {
              const Engine::Cherry & cc = e->cherries[child[x]];
              ok = (now[child[x]] || cc.valid) && cc.map_valid && cc.trackable && class_codes_match(e, child[x], lut_used) &&
                   child_scaler[x] == cc.scaler_index;          // (its counts per class are the ones asked for)
            }
          }
          if (ok)
          {
            bool failed = false;
            ok = class_map(e, o.parent_clv_index, child[0], child[1], lut_used, failed);
            if (failed) return PLL_FAILURE;
          }
          if (ok)
          {
            Engine::Cherry & c = e->cherries[o.parent_clv_index];
            if (!cherry_storage(e, o.parent_clv_index, c.nclasses)) return PLL_FAILURE;
            c.valid = true;
            c.materialized = false;
            c.scaler_index = o.parent_scaler_index;
            now[o.parent_clv_index] = 1;
            rp.cherry_ops.push_back(k);
          }
          else rp.ops.push_back(o);
        }
        if (!rp.cherry_ops.empty() && !rp.ops.empty())
        {
          rp.active = true;
          rp.ncodes = lut_used;
        }
        else
          for (unsigned k : rp.cherry_ops) e->cherries[ops[k].parent_clv_index].valid = false;
      }
      // (evaluate-only traversals: not together with site repeats, whose class operations leave the list)
      const bool transient = e->transient_mode && !e->site_repeats;
      const bool have = rp.active ? prepare_schedule(e, p, rp.ops.data(), (unsigned)rp.ops.size(), mode, &rp, ops, count)
                                  : prepare_schedule(e, p, ops, count, mode, nullptr, nullptr, 0, transient);
      if (!have && rp.active)
        for (unsigned k : rp.cherry_ops) e->cherries[ops[k].parent_clv_index].valid = false;     // the plain paths below compute them
      if (have)
      {
        PlanView view;
        if (!upload_plan(e->plan, e->stream, view)) return PLL_FAILURE;
        if (dp.ncherry_jobs || dp.npair_jobs)
        {
          if (!launch_class_levels(e, dp)) return PLL_FAILURE;
          // the scale buffers of the class operations hold their counts per class from now on (need_scaler)
          if (rp.active)
            for (unsigned k : rp.cherry_ops)
              if (ops[k].parent_scaler_index >= 0) e->scaler_lazy[ops[k].parent_scaler_index] = (int)ops[k].parent_clv_index;
          e->repeat_stats.cherries += dp.ncherry_jobs;
          e->repeat_stats.classes += dp.repeat_classes;
          e->repeat_stats.sites += (unsigned long long)dp.ncherry_jobs * e->N;
        }
        for (const DevicePlan::Launch & l : dp.launches)
        {
          const unsigned rows = l.rows;
          hipEvent_t ev1;
          if (!prof_begin(ev1)) return PLL_FAILURE;
          if (chains20 ? !launch_traverse_s20(e, view, dp.lds_doubles, dp.max_extent, l.begin, l.end, rows, 0, !e->cherries.empty(), transient)
                       : chains16 ? !launch_traverse_s16(e, view, dp.lds_doubles, dp.max_extent, l.begin, l.end, rows, 0, !e->cherries.empty())
                                  : !launch_traverse_s4(e, view, dp.lds_doubles, dp.max_extent, l.begin, l.end, rows, 0, transient, !e->cherries.empty()))
            return PLL_FAILURE;
          if (!prof_end(ev1, l.bytes, l.flops, l.ops, l.min_bytes)) return PLL_FAILURE;
          e->counters.partial_launches++;
        }
        if (transient) transient_after_list(e, ops, count, dp);
        e->counters.partial_ops += count;
        e->counters.site_updates += (unsigned long long)count * e->N * e->R;
        return PLL_SUCCESS;
      }
      plan = ChainPlan();
    }
    if (!plain_scalers()) return PLL_FAILURE;
    if (!chains16 && plan_chains(e, ops, count, chain_max, chain_lds, lut_used, plan))
    {
      for (int round = 0; round < plan.rounds; ++round)
      {
        ChainBatch cb;
        unsigned nops = 0, nchains = 0, longest = 0, lds_max = 0;
        double bytes = 0.0, flops = 0.0, minb = 0.0;
        auto flush = [&]() -> int
        {
          if (!nchains) return PLL_SUCCESS;
          hipEvent_t ev1;
          if (!prof_begin(ev1)) return PLL_FAILURE;
          if (longest == 1)
          {
            OpBatch ob;                       // nothing to hand over: the plain kernel
            for (unsigned i = 0; i < nops; ++i) ob.op[i] = cb.op[i];
            if (!launch_partials(e, ob, nops)) return PLL_FAILURE;
          }
          else if (chains20 ? !launch_chains_s20(e, cb, nchains, lds_max, lut_used)
                            : !launch_chains_s4(e, cb, nchains, longest))
            return PLL_FAILURE;
          if (!prof_end(ev1, bytes, flops, nops, bytes - minb)) return PLL_FAILURE;
          e->counters.partial_launches++;
          nops = nchains = longest = lds_max = 0;
          bytes = flops = minb = 0.0;
          return PLL_SUCCESS;
        };
        // longest chains first: their workgroups are dispatched first, which keeps the tail
        // of the launch short
        std::vector<size_t> order;
        for (size_t c = 0; c < plan.chains.size(); ++c)
          if (plan.launch[c] == round) order.push_back(c);
        std::stable_sort(order.begin(), order.end(),
                         [&](size_t a, size_t b) { return plan.chains[a].size() > plan.chains[b].size(); });
        for (size_t c : order)
        {
          const std::vector<unsigned> & ch = plan.chains[c];
          if (nops + ch.size() > MAX_OPS_PER_LAUNCH && !flush()) return PLL_FAILURE;
          cb.first[nchains] = (unsigned char)nops;
          cb.len[nchains] = (unsigned char)ch.size();
          unsigned off = 0;
          for (size_t i = 0; i < ch.size(); ++i)
          {
            const pll_operation_t & o = ops[ch[i]];
            if (!need_clv(e, o.child1_clv_index) || !need_clv(e, o.child2_clv_index)) return PLL_FAILURE;
            fill_desc(e, o, cb.op[nops], bytes, flops);
            cb.carried[nops] = i ? plan.carried[ch[i]] : 0;
            if (cb.carried[nops])       // handed over in registers: not read
              minb += (double)e->N * e->R * 8.0 * e->S +
                      ((cb.carried[nops] == 1 ? cb.op[nops].scaler1 : cb.op[nops].scaler2) ? 4.0 * (double)e->N * (e->rate_scalers ? e->R : 1) : 0.0);
            if (chains20)
            {
              const bool t1 = e->coded_tips && o.child1_clv_index < e->tips;
              const bool t2 = e->coded_tips && o.child2_clv_index < e->tips;
              cb.slot1[nops] = (unsigned short)off;
              off += s20_chain_slot(e, t1, lut_used);
              cb.slot2[nops] = (unsigned short)off;
              off += s20_chain_slot(e, t2, lut_used);
            }
            else cb.slot1[nops] = cb.slot2[nops] = 0;
            ++nops;
          }
          lds_max = std::max(lds_max, off);
          ++nchains;
          longest = std::max<unsigned>(longest, (unsigned)ch.size());
        }
        if (!flush()) return PLL_FAILURE;
      }
      e->counters.partial_ops += count;
      e->counters.site_updates += (unsigned long long)count * e->N * e->R;
      return PLL_SUCCESS;
    }
  }

  if (!plain_scalers()) return PLL_FAILURE;

  // 61 states.  (1) Cherries -- tip x tip operations, bound by HBM writes while the matrix cores
  // idle -- are folded into the operation that consumes them (kernels_s61.hpp, k_partials_s61v4).
  // (2) Those that stay (their consumer already folds its other child) run as LATE as the consumer
  // allows, so that they share a launch with matrix-bound operations.  Both only for lists with the
  // shape of a tree traversal (plan_chains' test); PLLHIP_S61_CHERRIES=0 / PLLHIP_S61_ALAP=0 switch
  // them off.
  static const int use_alap = getenv("PLLHIP_S61_ALAP") ? atoi(getenv("PLLHIP_S61_ALAP")) : 1;
  std::vector<char> light(count, 0);
  std::vector<int> cherry_of(count, -1);          // consumer -> the cherry folded into it
  std::vector<const uint8_t *> cherry_table(count, nullptr);
  bool folding = false;
  if (e->family == KernelFamily::S61 && lut_active(e) && count >= 2)
  {
    ChainPlan shape;
    if (plan_chains(e, ops, count, 1, ~0u, 1u, shape))
    {
      std::vector<int> producer(e->nodes, -1), consumer(count, -1), prod1(count, -1), prod2(count, -1);
      for (unsigned k = 0; k < count; ++k)
      {
        prod1[k] = producer[ops[k].child1_clv_index];
        prod2[k] = producer[ops[k].child2_clv_index];
        if (prod1[k] >= 0) consumer[prod1[k]] = (int)k;
        if (prod2[k] >= 0) consumer[prod2[k]] = (int)k;
        producer[ops[k].parent_clv_index] = (int)k;
      }
      auto is_cherry = [&](unsigned k) { return tip_coded(e, ops[k].child1_clv_index) && tip_coded(e, ops[k].child2_clv_index); };
      std::vector<char> folded(count, 0);
      if (s61_cherries_supported(e))
      {
        for (unsigned k = 0; k < count; ++k)
          if (is_cherry(k) && consumer[k] >= 0 && cherry_of[consumer[k]] < 0)
          {
            cherry_of[consumer[k]] = (int)k;
            folded[k] = 1;
            folding = true;
          }
      }
      if (folding)
      {
        // levels from the remaining dependencies alone (the list is a tree traversal: every vector
        // is written once and read by one later operation)
        max_level = 0;
        for (unsigned k = 0; k < count; ++k)
        {
          if (folded[k]) { level[k] = -1; continue; }
          int l = 0;
          if (prod1[k] >= 0 && !folded[prod1[k]]) l = std::max(l, level[prod1[k]] + 1);
          if (prod2[k] >= 0 && !folded[prod2[k]]) l = std::max(l, level[prod2[k]] + 1);
          level[k] = l;
          max_level = std::max(max_level, l);
        }
        // scaling decision of every folded cherry per pair of tip codes
        std::vector<unsigned> need;
        for (unsigned k = 0; k < count; ++k)
          if (folded[k] && ops[k].parent_scaler_index != PLL_SCALE_BUFFER_NONE) need.push_back(k);
        const size_t tab = (size_t)e->lut_codes * e->lut_codes;
        const size_t lut_stride = (size_t)e->R * e->lut_codes * e->S;
        if (need.size() * tab > e->s61_ttscale_cap)
        {
          PLLHIP_TRY(hipStreamSynchronize(e->stream));
          (void)hipFree(e->d_s61_ttscale);
          e->d_s61_ttscale = nullptr;
          e->s61_ttscale_cap = 0;
          if (!dev_alloc(&e->d_s61_ttscale, 2 * need.size() * tab, "cherry scaling tables")) return PLL_FAILURE;
          e->s61_ttscale_cap = 2 * need.size() * tab;
        }
        for (size_t i0 = 0; i0 < need.size(); i0 += 32)
        {
          CherryScaleBatch cs;
          memset(&cs, 0, sizeof(cs));
          const unsigned nc = (unsigned)std::min<size_t>(32, need.size() - i0);
          for (unsigned i = 0; i < nc; ++i)
          {
            const pll_operation_t & o = ops[need[i0 + i]];
            cs.lut1[i] = e->d_lut + lut_stride * o.child1_matrix_index;
            cs.lut2[i] = e->d_lut + lut_stride * o.child2_matrix_index;
            cs.out[i] = e->d_s61_ttscale + (i0 + i) * tab;
            cherry_table[need[i0 + i]] = cs.out[i];
          }
          if (!launch_cherry_scale_s61(e, cs, nc)) return PLL_FAILURE;
        }
      }
      if (use_alap)
        for (unsigned k = 0; k < count; ++k)
        {
          if (folded[k] || !is_cherry(k)) continue;
          if (consumer[k] >= 0 && level[consumer[k]] - 1 > level[k]) level[k] = level[consumer[k]] - 1;
          light[k] = 1;
        }
    }
  }

  for (int l = 0; l <= max_level; ++l)
  {
    OpBatch batch, cherries;
    const uint8_t * tables[MAX_OPS_PER_LAUNCH];
    unsigned nb = 0, nfolded = 0;
    bool any_cherry = false;
    double batch_bytes = 0.0, batch_flops = 0.0;
    const unsigned cap = folding ? S61_V4_OPS : MAX_OPS_PER_LAUNCH;
    // matrix-bound operations first, the light ones behind them (their workgroups fill the tail)
    for (unsigned pass = 0; pass < 2; ++pass)
    for (unsigned k = 0; k <= count; ++k)
    {
      if (k < count && (light[k] != (char)pass)) continue;
      if (k == count && pass == 0) continue;
      if (k < count && level[k] == l)
      {
        pll_operation_t o = ops[k];
        if (!need_clv(e, o.child1_clv_index) || !need_clv(e, o.child2_clv_index)) return PLL_FAILURE;
        memset(&cherries.op[nb], 0, sizeof(OpDesc));
        tables[nb] = nullptr;
        if (cherry_of[k] >= 0)
        {
          const pll_operation_t & c = ops[cherry_of[k]];
          if (o.child2_clv_index == c.parent_clv_index)      // the folded cherry is child 1 (products commute)
          {
            std::swap(o.child1_clv_index, o.child2_clv_index);
            std::swap(o.child1_matrix_index, o.child2_matrix_index);
            std::swap(o.child1_scaler_index, o.child2_scaler_index);
          }
          fill_desc(e, c, cherries.op[nb], batch_bytes, batch_flops);
          tables[nb] = cherry_table[cherry_of[k]];
          any_cherry = true;
          ++nfolded;
        }
        fill_desc(e, o, batch.op[nb++], batch_bytes, batch_flops);
      }
      if (nb == cap || (k == count && nb))
      {
        hipEvent_t ev1;
        if (!prof_begin(ev1)) return PLL_FAILURE;
        // (with cherries folded somewhere in the traversal, every launch takes the 512-thread kernel: 0.5 - 3 % faster
        // than switching to the 256-thread one for launches without a cherry)
        static const int all_v4 = getenv("PLLHIP_S61_ALLV4") ? atoi(getenv("PLLHIP_S61_ALLV4")) : 1;
        if ((any_cherry || (all_v4 && folding)) ? !launch_partials_s61_cherries(e, batch, cherries, tables, nb) : !launch_partials(e, batch, nb))
          return PLL_FAILURE;
        if (!prof_end(ev1, batch_bytes, batch_flops, nb + nfolded)) return PLL_FAILURE;
        e->counters.partial_launches++;
        nb = nfolded = 0;
        any_cherry = false;
        batch_bytes = 0.0;
        batch_flops = 0.0;
      }
    }
  }
  e->counters.partial_ops += count;
  e->counters.site_updates += (unsigned long long)count * e->N * e->R;
  return PLL_SUCCESS;
}

} // namespace pllhip


namespace pllhip {

// ---------------------------------------------------------------------------
// One operation list over several partitions (pllhip_update_partials_batch).
// pll-modules evaluates every partition of an analysis on the same tree, one after the other
// (src/tree/treeinfo.c:1020-1056); with one partition per gene a partition holds a few thousand sites
// and every launch of its own costs more than it computes.  Partitions of one kernel family on one
// device share their launches instead: the round schedules of the members (prepare_schedule) are laid
// side by side in ONE resident schedule -- the chains of a round of every member become grid rows of the
// same launch, a chain knows its partition's extent and tables (PlanChain) -- on the first member's stream,
// with one event per member on either side.  What is stored is what the per-partition calls store,
// bit for bit: the same kernels run the same chains.
// ---------------------------------------------------------------------------
static bool batch_family(const Engine * e)
{
  static const int use_chains = getenv("PLLHIP_CHAINS") ? atoi(getenv("PLLHIP_CHAINS")) : 1;
  static const int use_batch = getenv("PLLHIP_BATCH") ? atoi(getenv("PLLHIP_BATCH")) : 1;
  if (!use_chains || !use_batch || !e->shards.empty() || e->site_repeats) return false;
  return (e->family == KernelFamily::S20 && chains_supported_s20(e)) ||
         (e->family == KernelFamily::S4 && chains_supported_s4(e)) ||
         (e->family == KernelFamily::S16 && chains_supported_s16(e));
}

static bool batch_compatible(const Engine * a, const Engine * b)
{
  return a->device == b->device && a->family == b->family && a->S == b->S && a->R == b->R &&
         a->rate_scalers == b->rate_scalers;
}

static void batch_free(BatchPlan * b)
{
  if (!b) return;
  (void)hipFree(b->plan.d_buf);
  if (b->plan.h_stage) (void)hipHostFree(b->plan.h_stage);
  if (b->plan.copied) (void)hipEventDestroy(b->plan.copied);
  for (hipEvent_t ev : b->ready) if (ev) (void)hipEventDestroy(ev);
  if (b->done) (void)hipEventDestroy(b->done);
  delete b;
}

// 1 = done as one batch, 0 = the members' schedules cannot share launches (the caller falls back to
// per-partition calls), -1 = error
static int update_partials_group(const std::vector<pll_partition_t *> & g, const pll_operation_t * ops, unsigned count)
{
  Engine * lead = engine_of(g[0]);
  const size_t M = g.size();
  if (hipSetDevice(lead->device) != hipSuccess) { set_error(PLL_ERROR_HIP_RUNTIME, "hipSetDevice"); return -1; }
  // One launch for the whole list (every member gets grid rows of its own, whose workgroups walk ALL chains
  // of the member for their share of its site blocks: no launch boundary, no tail between rounds), or one
  // launch per round of chains (the chains of a round of every member side by side)?  The first needs enough
  // site blocks in the group to fill the chip without the parallelism of the chains.  PLLHIP_BATCH_MODE=1 / 0.
  static const int env_mode = getenv("PLLHIP_BATCH_MODE") ? atoi(getenv("PLLHIP_BATCH_MODE")) : -1;
  unsigned long long units = 0;       // workgroups the members could use side by side
  for (pll_partition_t * p : g)
  {
    const Engine * e = engine_of(p);
    units += e->family == KernelFamily::S4 ? (e->N + 255u) / 256u : (e->nblk + 7u) / 8u;
  }
  const unsigned mode = env_mode >= 0 ? (env_mode ? 1u : 0u) : (units >= 2ull * lead->cu_count && count >= 2) ? 1u : 0u;
  for (pll_partition_t * p : g)
  {
    Engine * e = engine_of(p);
    if (!validate_ops(e, ops, count)) return -1;
    if (e->ntransient && (hipSetDevice(e->device) != hipSuccess || !transient_before_list(e, ops, count))) return -1;
    // (tips kept per class are written off as class nodes by an operation that overwrote one: never -- tips are not
    // parents; the vectors this list writes stop being whatever class node they were)
    if (!e->cherries.empty())
      for (unsigned k = 0; k < count; ++k) e->cherries[ops[k].parent_clv_index].valid = false;
    if (!prepare_schedule(e, p, ops, count, mode, nullptr, nullptr, 0, e->transient_mode && !e->site_repeats)) return 0;
  }
  // launches can be shared when every member's schedule has the same shape (same list, same family: always,
  // unless the tables of one member fill the LDS earlier and cut its chains elsewhere)
  const DevicePlan & first = lead->plan;
  for (size_t m = 1; m < M; ++m)
  {
    const DevicePlan & dp = engine_of(g[m])->plan;
    if (dp.launches.size() != first.launches.size() || dp.nchains != first.nchains) return 0;
    for (size_t j = 0; j < dp.launches.size(); ++j)
      if (dp.launches[j].rows != first.launches[j].rows ||
          dp.launches[j].end - dp.launches[j].begin != first.launches[j].end - first.launches[j].begin)
        return 0;
  }

  if (!lead->batch) lead->batch = new BatchPlan();
  BatchPlan & b = *lead->batch;
  bool same = b.members.size() == M;
  for (size_t m = 0; same && m < M; ++m)
    same = b.members[m] == (const void *)engine_of(g[m]) && b.generations[m] == engine_of(g[m])->plan.generation;
  if (!same)
  {
    b.members.resize(M);
    b.generations.resize(M);
    DevicePlan & mp = b.plan;
    std::vector<PlanOp> pops;
    std::vector<PlanChain> pchains;
    std::vector<unsigned> base(M);
    mp.lds_doubles = 0;
    mp.max_extent = 0;
    for (size_t m = 0; m < M; ++m)
    {
      const DevicePlan & dp = engine_of(g[m])->plan;
      b.members[m] = engine_of(g[m]);
      b.generations[m] = dp.generation;
      base[m] = (unsigned)pops.size();
      const PlanOp * src = reinterpret_cast<const PlanOp *>(dp.bytes.data());
      pops.insert(pops.end(), src, src + dp.nops);
      mp.lds_doubles = std::max(mp.lds_doubles, dp.lds_doubles);
      mp.max_extent = std::max(mp.max_extent, dp.max_extent);
    }
    mp.launches.clear();
    for (size_t j = 0; j < first.launches.size(); ++j)
    {
      DevicePlan::Launch l = first.launches[j];
      const unsigned nch = l.end - l.begin;
      l.begin = (unsigned)pchains.size();
      l.ops = 0;
      l.bytes = l.flops = l.min_bytes = 0.0;
      // a round: the chains of every member side by side (a row each).  A run of single chains (rows == 1:
      // one row walks them in turn): step-major, so that with one row per member row m walks member m's chains
      const bool run = first.launches[j].rows == 1;
      const size_t row0 = pchains.size();
      for (unsigned x = 0; x < (run ? nch : (unsigned)M); ++x)
        for (unsigned y = 0; y < (run ? (unsigned)M : nch); ++y)
        {
          const size_t m = run ? y : x;
          const unsigned c = run ? x : y;
          const DevicePlan & dp = engine_of(g[m])->plan;
          const PlanChain * chains = reinterpret_cast<const PlanChain *>(dp.bytes.data() + (size_t)dp.nops * sizeof(PlanOp));
          PlanChain pc = chains[dp.launches[j].begin + c];
          pc.first += base[m];
          pchains.push_back(pc);
        }
      // the rows of a round are independent: the most expensive ones (operations x site blocks) are dispatched first,
      // whatever partition they belong to, which keeps the tail of the launch short
      if (!run)
        std::stable_sort(pchains.begin() + row0, pchains.end(), [](const PlanChain & a, const PlanChain & b)
        { return (unsigned long long)a.len * a.extent > (unsigned long long)b.len * b.extent; });
      for (size_t m = 0; m < M; ++m)
      {
        const DevicePlan::Launch & lm = engine_of(g[m])->plan.launches[j];
        l.ops += lm.ops; l.bytes += lm.bytes; l.flops += lm.flops; l.min_bytes += lm.min_bytes;
      }
      l.end = (unsigned)pchains.size();
      l.rows = first.launches[j].rows * (unsigned)M;
      mp.launches.push_back(l);
    }
    mp.nops = (unsigned)pops.size();
    mp.nchains = (unsigned)pchains.size();
    mp.bytes.resize(pops.size() * sizeof(PlanOp) + pchains.size() * sizeof(PlanChain));
    memcpy(mp.bytes.data(), pops.data(), pops.size() * sizeof(PlanOp));
    memcpy(mp.bytes.data() + pops.size() * sizeof(PlanOp), pchains.data(), pchains.size() * sizeof(PlanChain));
  }
  while (b.ready.size() < M)
  {
    hipEvent_t ev;
    if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return -1;
    b.ready.push_back(ev);
  }
  if (!b.done && !hip_ok(hipEventCreateWithFlags(&b.done, hipEventDisableTiming), "hipEventCreate")) return -1;

  // the row tables of the members' wide tips (tips kept per class of sites), each on its member's stream
  bool any_wide = false;
  for (pll_partition_t * p : g)
  {
    Engine * e = engine_of(p);
    any_wide = any_wide || !e->cherries.empty();
    if (!e->plan.npair_jobs) continue;
    PlanView own;
    if (hipSetDevice(e->device) != hipSuccess || !upload_plan(e->plan, e->stream, own) || !launch_class_levels(e, e->plan)) return -1;
  }
  // the members' streams have issued what the traversal reads (P-matrices, tables, tip data)
  for (size_t m = 1; m < M; ++m)
    if (!hip_ok(hipEventRecord(b.ready[m], engine_of(g[m])->stream), "hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(lead->stream, b.ready[m], 0), "hipStreamWaitEvent"))
      return -1;
  PlanView view;
  if (!upload_plan(b.plan, lead->stream, view)) return -1;
  for (const DevicePlan::Launch & l : b.plan.launches)
  {
    hipEvent_t ev1 = nullptr;
    if (lead->profiling)
    {
      if (lead->prof_used == lead->prof_events.size())
      {
        hipEvent_t x, y;
        if (!hip_ok(hipEventCreate(&x), "hipEventCreate") || !hip_ok(hipEventCreate(&y), "hipEventCreate")) return -1;
        lead->prof_events.emplace_back(x, y);
      }
      ev1 = lead->prof_events[lead->prof_used].second;
      if (!hip_ok(hipEventRecord(lead->prof_events[lead->prof_used].first, lead->stream), "hipEventRecord")) return -1;
      lead->prof_used++;
    }
    // one launch for the whole list: the rows are the members, every workgroup has the same long walk in front
    // of it, so exactly the workgroups that are resident at once (one per CU; three at 4 states)
    static const int env_walk = getenv("PLLHIP_BATCH_WGS") ? atoi(getenv("PLLHIP_BATCH_WGS")) : 0;
    const unsigned wgs = mode ? (env_walk > 0 ? (unsigned)env_walk : lead->family == KernelFamily::S4 ? 3u : 1u) : 0u;
    bool any_transient = false;
    for (pll_partition_t * p : g) any_transient = any_transient || (engine_of(p)->transient_mode && !engine_of(p)->site_repeats);
    const int ok = lead->family == KernelFamily::S20 ? launch_traverse_s20(lead, view, b.plan.lds_doubles, b.plan.max_extent, l.begin, l.end, l.rows, wgs, any_wide, any_transient)
                 : lead->family == KernelFamily::S16 ? launch_traverse_s16(lead, view, b.plan.lds_doubles, b.plan.max_extent, l.begin, l.end, l.rows, wgs, any_wide)
                                                     : launch_traverse_s4(lead, view, b.plan.lds_doubles, b.plan.max_extent, l.begin, l.end, l.rows, wgs, any_transient, any_wide);
    if (!ok) return -1;
    if (lead->profiling)
    {
      if (!hip_ok(hipEventRecord(ev1, lead->stream), "hipEventRecord")) return -1;
      lead->prof_bytes += l.bytes;
      lead->prof_min_bytes += l.min_bytes;
      lead->prof_flops += l.flops;
      lead->prof_ops += l.ops;
    }
    lead->counters.partial_launches++;
  }
  // whatever a member's stream does next comes after the traversal
  if (!hip_ok(hipEventRecord(b.done, lead->stream), "hipEventRecord")) return -1;
  for (size_t m = 1; m < M; ++m)
    if (!hip_ok(hipStreamWaitEvent(engine_of(g[m])->stream, b.done, 0), "hipStreamWaitEvent")) return -1;
  for (pll_partition_t * p : g)
  {
    Engine * e = engine_of(p);
    if (e->transient_mode && !e->site_repeats) transient_after_list(e, ops, count, e->plan);
    e->counters.partial_ops += count;
    e->counters.site_updates += (unsigned long long)count * e->N * e->R;
  }
  return 1;
}

} // namespace pllhip

extern "C" int pllhip_update_partials_batch(pll_partition_t * const * partitions, unsigned int partition_count,
                                            const pll_operation_t * ops, unsigned int count)
{
  using namespace pllhip;
  if (!count) return PLL_SUCCESS;
  int rc = PLL_SUCCESS;
  std::vector<char> taken(partition_count, 0);
  // pending P-matrices and the tip tables first: whether a partition's traversals can run from a shared
  // schedule depends on its tables (and every member needs them anyway)
  for (unsigned i = 0; i < partition_count; ++i)
    if (partitions[i] && !is_router(partitions[i]))
    {
      Engine * e = engine_of(partitions[i]);
      PLLHIP_TRY(hipSetDevice(e->device));
      // (vectors an evaluate-only traversal did not store: settled before the queued P-matrices are launched)
      if (e->ntransient && (!validate_ops(e, ops, count) || !transient_before_list(e, ops, count))) return PLL_FAILURE;
      if (!flush_pmatrices(partitions[i]) || !ensure_luts(partitions[i])) return PLL_FAILURE;
    }
  for (unsigned i = 0; i < partition_count; ++i)
  {
    if (!partitions[i] || taken[i]) continue;
    taken[i] = 1;
    std::vector<pll_partition_t *> group(1, partitions[i]);
    const Engine * e = engine_of(partitions[i]);
    if (batch_family(e))
      for (unsigned j = i + 1; j < partition_count; ++j)
        if (partitions[j] && !taken[j] && partitions[j] != partitions[i] && batch_family(engine_of(partitions[j])) &&
            batch_compatible(e, engine_of(partitions[j])))
        {
          taken[j] = 1;
          group.push_back(partitions[j]);
        }
    int done = 0;
    if (group.size() > 1)
    {
      done = update_partials_group(group, ops, count);
      if (done < 0) { rc = PLL_FAILURE; continue; }
    }
    if (!done)
      for (pll_partition_t * p : group)
      {
        pll_errno = 0;
        pll_update_partials(p, ops, count);
        if (pll_errno) rc = PLL_FAILURE;
      }
  }
  return rc;
}

namespace pllhip {

// ---------------------------------------------------------------------------
// router: a partition spread over several devices (Engine::shards)
// ---------------------------------------------------------------------------
// where a shard's next reduction leaves its totals: its own mapped result buffer
static Engine::Sink shard_sink(Engine * c)
{
  Engine::Sink sk;
  sk.dst = c->d_result;
  sk.flag = reinterpret_cast<unsigned long long *>(c->d_result) + RESULT_SEQ_SLOT;
  sk.seq = ++c->result_seq;
  sk.nq = 0;
  return sk;
}

static int shard_wait(Engine * c, unsigned long long seq)
{
  const volatile unsigned long long * flag =
      reinterpret_cast<const volatile unsigned long long *>(c->h_result) + RESULT_SEQ_SLOT;
  return wait_sequence(c->stream, flag, seq);
}

// every shard's kernel is enqueued (all devices run at once), then the totals are collected and
// added in shard order: bit-reproducible, whatever the devices' finishing order
static double router_loglikelihood(pll_partition_t * p, unsigned pc, int psc, unsigned cc, int csc,
                                   int matrix_index, const unsigned * freqs_indices, double * persite_lnl)
{
  Engine * r = engine_of(p);
  const double fail = -std::numeric_limits<double>::infinity();
  std::vector<unsigned long long> seq(r->shards.size());
  for (size_t k = 0; k < r->shards.size(); ++k)
  {
    pll_partition_t * c = r->shards[k];
    push_model(p, c);
    Engine::Sink sk = shard_sink(engine_of(c));
    seq[k] = sk.seq;
    const double v = loglikelihood_impl(c, pc, psc, cc, csc, matrix_index, freqs_indices,
                                        persite_lnl ? persite_lnl + r->shard_first[k] : nullptr, &sk);
    if (v != 0.0) return fail;
  }
  double total = 0.0;
  for (size_t k = 0; k < r->shards.size(); ++k)
  {
    Engine * c = engine_of(r->shards[k]);
    if (hipSetDevice(c->device) != hipSuccess || !shard_wait(c, seq[k])) return fail;
    if (persite_lnl && !hip_ok(hipStreamSynchronize(c->stream), "persite sync")) return fail;
    total += c->h_result[0];
  }
  return total;
}

static int router_derivatives(pll_partition_t * p, int psc, int csc, const double * brlens, unsigned count,
                              const unsigned * params_indices, const double * sumtable,
                              double * out_df, double * out_ddf)
{
  Engine * r = engine_of(p);
  std::vector<unsigned long long> seq(r->shards.size());
  for (size_t k = 0; k < r->shards.size(); ++k)
  {
    pll_partition_t * c = r->shards[k];
    push_model(p, c);
    Engine::Sink sk = shard_sink(engine_of(c));
    seq[k] = sk.seq;
    if (!derivatives_impl(c, psc, csc, brlens, count, params_indices, sumtable, &sk, nullptr, nullptr))
      return PLL_FAILURE;
  }
  for (unsigned i = 0; i < count; ++i) out_df[i] = out_ddf[i] = 0.0;
  for (size_t k = 0; k < r->shards.size(); ++k)
  {
    Engine * c = engine_of(r->shards[k]);
    PLLHIP_TRY(hipSetDevice(c->device));
    if (!shard_wait(c, seq[k])) return PLL_FAILURE;
    for (unsigned i = 0; i < count; ++i) { out_df[i] += c->h_result[2 * i]; out_ddf[i] += c->h_result[2 * i + 1]; }
  }
  return PLL_SUCCESS;
}

} // namespace pllhip

using namespace pllhip;

// ===========================================================================
// B0 entry points
// ===========================================================================
extern "C" {

int pll_update_prob_matrices(pll_partition_t * p,
                             const unsigned int * params_indices,
                             const unsigned int * matrix_indices,
                             const double * branch_lengths,
                             unsigned int count)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    // the decomposition runs once, on the parent's arrays; every shard builds the (tiny) matrices itself
    if (!ensure_eigen(p, params_indices)) return PLL_FAILURE;
    for (pll_partition_t * c : e->shards)
    {
      push_model(p, c);
      if (!pll_update_prob_matrices(c, params_indices, matrix_indices, branch_lengths, count)) return PLL_FAILURE;
    }
    return PLL_SUCCESS;
  }
  // (no HIP call on the queueing path: the reference issues one call per branch and partition -- 6 304 per
  // evaluation of 32 partitions on 100 taxa --; the device is selected where something is launched or copied)
  if (!ensure_eigen(p, params_indices)) return PLL_FAILURE;
  if (!sync_model(p, e->pmatrix_burst)) return PLL_FAILURE;
  e->pmatrix_burst = true;                    // until another entry point looks at the model
  for (unsigned m = 0; m < count; ++m)
    if (matrix_indices[m] >= e->nmat || !(branch_lengths[m] >= 0.0))
    {
      set_error(PLL_ERROR_PARAM_INVALID, "Invalid matrix index %u or branch length %g",
                matrix_indices[m], branch_lengths[m]);
      return PLL_FAILURE;
    }
  // queue; a request for a matrix that is already queued replaces it
  const ParamIdx params = make_params(p, params_indices);
  if (!e->pend_midx.empty() && memcmp(&params, &e->pend_params, sizeof(params)) != 0)
    if (!flush_pmatrices(p)) return PLL_FAILURE;
  e->pend_params = params;
  if (e->pend_pos.size() != e->nmat) e->pend_pos.assign(e->nmat, -1);
  for (unsigned m = 0; m < count; ++m)
  {
    const unsigned idx = matrix_indices[m];
    if (e->pend_pos[idx] >= 0)
      e->pend_t[e->pend_pos[idx]] = branch_lengths[m];
    else
    {
      e->pend_pos[idx] = (int)e->pend_midx.size();
      e->pend_midx.push_back(idx);
      e->pend_t.push_back(branch_lengths[m]);
    }
  }
  e->pmat_host_dirty = true;
  e->counters.pmatrix_updates += count;
  // every matrix of the partition is queued: no later request can join this launch (a repeated one replaces its
  // entry), so it goes now -- with several partitions the kernel of this one runs while the host issues the
  // per-branch calls of the next (src/tree/treeinfo.c:845-865 loops the partitions inside every branch)
  // (not while vectors exist as their operation only: the launch waits for the consumer -- if that is the next
  // traversal, it gives those vectors up before the matrices they were made with change; flush_pmatrices)
  if (!e->ntransient && (e->pend_midx.size() >= 4 * MAX_PMAT_PER_LAUNCH || e->pend_midx.size() == e->nmat)) return flush_pmatrices(p);
  return PLL_SUCCESS;
}

void pll_update_partials(pll_partition_t * p, const pll_operation_t * ops, unsigned int count)
{
  if (!count) return;
  // void in the reference interface: errors are reported through pll_errno
  if (is_router(p))
  {
    for (pll_partition_t * c : engine_of(p)->shards) (void)update_partials_impl(c, ops, count);
    return;
  }
  (void)update_partials_impl(p, ops, count);
}

} // extern "C"

namespace pllhip {

// lnL at an edge (matrix_index >= 0) or at a root vector.  `deferred` == nullptr: the
// value comes back (one wait); otherwise the total is left at deferred->dst (a device
// slot or mapped memory, with deferred->flag / seq if the host is to be told), nothing
// is waited for and the return value is 0.
double loglikelihood_impl(pll_partition_t * p, unsigned pc, int psc, unsigned cc, int csc,
                          int matrix_index, const unsigned * freqs_indices,
                          double * persite_lnl, const Engine::Sink * deferred)
{
  Engine * e = engine_of(p);
  const double fail = -std::numeric_limits<double>::infinity();
  if (hipSetDevice(e->device) != hipSuccess) { set_error(PLL_ERROR_HIP_RUNTIME, "hipSetDevice"); return fail; }
  if (!check_clv_index(e, pc, "parent") || !check_scaler_index(e, psc)) return fail;
  if (matrix_index >= 0)
  {
    if (!check_clv_index(e, cc, "child") || !check_scaler_index(e, csc)) return fail;
    if ((unsigned)matrix_index >= e->nmat) { set_error(PLL_ERROR_PARAM_INVALID, "matrix index out of range"); return fail; }
  }
  if (!sync_model(p) || !flush_pmatrices(p) || !ensure_luts(p) || !ensure_invariant(p)) return fail;
  if (!need_clv(e, pc) || (matrix_index >= 0 && !need_clv(e, cc))) return fail;
  if (!need_scaler(e, psc) || (matrix_index >= 0 && !need_scaler(e, csc))) return fail;
  // ascertainment-bias correction: the kernel runs over alignment + constant patterns (the
  // latter weigh 0 on the device), the per-site values of the constant patterns come back
  // through the mapped result buffer and the host adds the closed-form correction
  const bool asc = e->N > e->Nreal;
  if (asc && deferred)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "internal: deferred lnL of an AB partition goes through the blocking form");
    return fail;
  }
  double * const ps_out_req = persite_lnl;
  if ((persite_lnl || asc) && !e->d_persite)
    if (!dev_alloc(&e->d_persite, (size_t)e->N, "per-site lnL")) return fail;
  if (asc && !persite_lnl) persite_lnl = e->h_asc;      // any non-null value: "per-site output on"

  const unsigned nblocks = reduce_grid(e);
  const ModelView mv = model_view(e);
  const ParamIdx fidx = make_params(p, freqs_indices);
  const NodeRef parent = node_ref(e, pc);
  if (deferred) e->sink = *deferred; else sink_to_host(e);
  e->sink.nq = 1;
  unsigned long long * const host_flag = e->sink.flag;
  if (asc) e->sink.flag = nullptr;                      // k_publish_tail tells the host, after the tail
  NodeRef child = {nullptr, nullptr};
  const double * pm = nullptr, * lut = nullptr;
  if (matrix_index >= 0)
  {
    child = node_ref(e, cc);
    pm = e->d_pmat + (size_t)matrix_index * e->R * e->S * e->Sp;
    if (child.codes) lut = e->d_lut + (size_t)matrix_index * e->R * e->lut_codes * e->S;
  }
  int rc;
  if (e->family == KernelFamily::S4 && matrix_index >= 0)
    rc = launch_edge_lnl_s4(e, mv, fidx, parent, child, pm, lut, scaler_ptr(e, psc), scaler_ptr(e, csc),
                            persite_lnl ? e->d_persite : nullptr, nblocks);
  else if (e->family == KernelFamily::S20)
    rc = launch_edge_lnl_s20(e, mv, fidx, parent, child, pm, lut, scaler_ptr(e, psc), scaler_ptr(e, csc),
                             persite_lnl ? e->d_persite : nullptr, nblocks);
  else if (e->family == KernelFamily::S61)
    rc = launch_edge_lnl_s61(e, mv, fidx, parent, child, pm, lut, scaler_ptr(e, psc), scaler_ptr(e, csc),
                             persite_lnl ? e->d_persite : nullptr, nblocks);
  else if (e->family == KernelFamily::S16)
    rc = launch_edge_lnl_s16(e, mv, fidx, parent, child, pm, lut, scaler_ptr(e, psc),
                             (matrix_index >= 0) ? scaler_ptr(e, csc) : nullptr,
                             persite_lnl ? e->d_persite : nullptr, nblocks);
  else
    rc = launch_edge_lnl_generic(e, mv, fidx, parent, child, pm, lut, scaler_ptr(e, psc),
                                 (matrix_index >= 0) ? scaler_ptr(e, csc) : nullptr,
                                 persite_lnl ? e->d_persite : nullptr, nblocks);
  if (!rc) return fail;
  double total = 0.0;
  e->counters.lnl_calls++;
  if (deferred)
  {
    if (!finish_launch(e, nblocks, 1)) return fail;
    // (the caller waits for the stream before it reads a per-site buffer)
    if (ps_out_req && e->Nreal &&
        !hip_ok(hipMemcpyAsync(ps_out_req, e->d_persite, sizeof(double) * e->Nreal, hipMemcpyDeviceToHost, e->stream),
                "persite copy")) return fail;
    return 0.0;
  }
  if (asc)
  {
    if (!finish_launch(e, nblocks, 1)) return fail;
    hipLaunchKernelGGL(k_publish_tail, dim3(1), dim3(64), 0, e->stream, e->d_persite + e->Nreal,
                       e->d_result + RESULT_ASC_SLOT, e->S, host_flag, e->sink.seq);
    if (!hip_ok(hipGetLastError(), "publish tail")) return fail;
    const volatile unsigned long long * flag =
        reinterpret_cast<const volatile unsigned long long *>(e->h_result) + RESULT_SEQ_SLOT;
    if (!wait_sequence(e->stream, flag, e->sink.seq)) return fail;
    total = e->h_result[0] + asc_correction(p, e->h_result + RESULT_ASC_SLOT);
  }
  else if (!finish_reduction(e, nblocks, 1, &total)) return fail;
  if (ps_out_req && e->Nreal)
  {
    // only the alignment patterns go to the caller; wait for the copy (the result flag came first)
    if (!hip_ok(hipMemcpyAsync(ps_out_req, e->d_persite, sizeof(double) * e->Nreal,
                               hipMemcpyDeviceToHost, e->stream), "persite copy") ||
        !hip_ok(hipStreamSynchronize(e->stream), "persite sync")) return fail;
  }
  return total;
}

// the loop kernel of pllhip_newton_branch for this partition's family, its LDS, and how many of its workgroups
// the chip holds at once (-1: HIP error).  derivatives_impl sizes its scan grid with the same number, so that the
// device loop and the host loop add the same block totals in the same order.
static int newton_capacity(Engine * e, const void ** fn_out, size_t * lds_out)
{
  if (e->family == KernelFamily::S4)
  {
    // 4 states (kernels_newton_s4.hpp): the loop with the table in registers when a thread has at most two trips
    // on the grid that variant can hold at once; otherwise the host loop (capacity 0: a streaming loop in one
    // launch measured no faster than one blocking call per iterate -- 59.3 against 57.7 us all-in at 1 M sites)
    if (e->newton_capacity < 0)
    {
      static const int env_res = getenv("PLLHIP_NEWTON_RESIDENT") ? atoi(getenv("PLLHIP_NEWTON_RESIDENT")) : 1;
      const void * fn2 = reinterpret_cast<const void *>(k_newton_s4<2>);
      int cu2 = 0;
      if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&cu2, fn2, 256, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor"))
        return -1;
      const unsigned cap2 = (unsigned)std::max(0, cu2) * e->cu_count;
      const unsigned long long limit = ((unsigned long long)e->N * e->R + 63ULL) & ~63ULL;
      // (the grid the loop is launched on: scan_grid() = the reduction grid, at most four workgroups per CU and at
      // most what the chip holds at once)
      const unsigned g2 = std::min({reduce_grid(e), 4u * e->cu_count, cap2});
      const bool resident = env_res && g2 && (limit + 1024ULL * g2 - 1) / (1024ULL * g2) <= 2;
      e->newton_resident = resident ? 2 : 0;
      e->newton_fn = fn2;
      e->newton_lds = 0;
      e->newton_capacity = resident ? (int)cap2 : 0;
    }
    if (fn_out) *fn_out = e->newton_fn;
    if (lds_out) *lds_out = 0;
    return e->newton_capacity;
  }
  const unsigned ks = e->family == KernelFamily::S20 ? 5u : e->family == KernelFamily::S61 ? S61_KS : s16_ks(e);
  const size_t lds = sizeof(double) * e->R * ks * 64;
  const void * fn = nullptr;
#define PLLHIP_PICK(KK, SS) fn = reinterpret_cast<const void *>(k_newton_mfma<KK, SS>)
  if (e->family == KernelFamily::S20) PLLHIP_PICK(5, 20);
  else if (e->family == KernelFamily::S61) { if (e->S == S61_S) PLLHIP_PICK(S61_KS, S61_S); else PLLHIP_PICK(S61_KS, 0); }
  else
    switch (ks)
    {
      case 1: PLLHIP_PICK(1, 0); break;
      case 2: PLLHIP_PICK(2, 0); break;
      case 3: PLLHIP_PICK(3, 0); break;
      case 4: PLLHIP_PICK(4, 0); break;
      case 5: PLLHIP_PICK(5, 0); break;
      case 6: PLLHIP_PICK(6, 0); break;
      case 7: PLLHIP_PICK(7, 0); break;
      default: PLLHIP_PICK(8, 0); break;
    }
#undef PLLHIP_PICK
  if (lds_out) *lds_out = lds;
  if (e->newton_capacity < 0)
  {
    if (lds > 160 * 1024 - 512) { e->newton_capacity = 0; return 0; }
    if (lds > 64 * 1024 && !hip_ok(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512), "hipFuncSetAttribute"))
      return -1;
    int per_cu = 0;
    if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor"))
      return -1;
    e->newton_capacity = std::max(0, per_cu) * (int)e->cu_count;
    e->newton_stream_fn = fn;
    e->newton_stream_lds = lds;
    e->newton_stream_capacity = e->newton_capacity;
    // Does the sumtable fit the registers of the waves of ONE co-resident grid (k_newton_mfma_resident)?  Then that
    // grid is the scan grid of this partition -- for the single scans of derivatives_impl as well, so that both
    // add the same block totals in the same order.
    static const int env_res = getenv("PLLHIP_NEWTON_RESIDENT") ? atoi(getenv("PLLHIP_NEWTON_RESIDENT")) : 1;
    e->newton_resident = 0;
    e->newton_fn = nullptr;
    const void * rfn = nullptr;
    unsigned nb = 0;
    size_t res_lds = lds;
    if (env_res && e->R == 4 && !e->rate_scalers && lds <= 64 * 1024)
    {
      if (e->family == KernelFamily::S20) { rfn = reinterpret_cast<const void *>(k_newton_mfma_resident<5, 20, 4, 3>); nb = 4; res_lds = lds + sizeof(double2) * 4 * 4 * 5 * 64; }
      else if (e->family == KernelFamily::S61)
      {
        rfn = e->S == S61_S ? reinterpret_cast<const void *>(k_newton_mfma_resident<S61_KS, S61_S, 1, 1>)
                            : reinterpret_cast<const void *>(k_newton_mfma_resident<S61_KS, 0, 1, 1>);
        nb = 1;
      }
      else if (e->family == KernelFamily::S16)
      {
        // 2 .. 32 states: as many blocks in registers as ~240 of them hold (16 KS per block), one more in LDS
#define PLLHIP_RES(KK, NBR) do { rfn = reinterpret_cast<const void *>(k_newton_mfma_resident<KK, 0, NBR + 1, NBR>); nb = NBR + 1; \
                                 res_lds = lds + sizeof(double2) * 4 * 4 * KK * 64; } while (0)
        switch (ks)
        {
          case 1: PLLHIP_RES(1, 7); break;
          case 2: PLLHIP_RES(2, 7); break;
          case 3: PLLHIP_RES(3, 5); break;
          case 4: PLLHIP_RES(4, 3); break;
          case 5: PLLHIP_RES(5, 3); break;
          case 6: PLLHIP_RES(6, 2); break;
          case 7: PLLHIP_RES(7, 2); break;
          default: PLLHIP_RES(8, 1); break;
        }
#undef PLLHIP_RES
      }
    }
    if (rfn)
    {
      int res_cu = 0;
      if (res_lds > 64 * 1024 && !hip_ok(hipFuncSetAttribute(rfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)res_lds), "hipFuncSetAttribute"))
        return -1;
      if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&res_cu, rfn, 256, res_lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor"))
        return -1;
      const unsigned cap_res = (unsigned)std::max(0, res_cu) * e->cu_count;
      const unsigned g = std::min({reduce_grid(e), 4u * e->cu_count, cap_res});
      if (g && (unsigned long long)e->nblk <= (unsigned long long)nb * 4u * g)
      {
        e->newton_resident = (int)nb;
        e->newton_fn = rfn;
        e->newton_lds = res_lds;
        e->newton_capacity = (int)cap_res;
      }
    }
  }
  if (fn_out) *fn_out = e->newton_resident ? e->newton_fn : fn;
  if (lds_out && e->newton_resident) *lds_out = e->newton_lds;
  return e->newton_capacity;
}

// workgroups of a derivative scan on the matrix cores: long-lived ones (the scan pipelines its loads across the
// units a wave walks; tables built, totals reduced and published less often), and no more than the chip holds at
// once, so that pllhip_newton_branch can run the very same grid as a loop
static unsigned scan_grid(Engine * e)
{
  unsigned nblocks = std::min(reduce_grid(e), 4u * e->cu_count);
  const int cap = newton_capacity(e, nullptr, nullptr);
  if (cap > 0) nblocks = std::min(nblocks, (unsigned)cap);
  return nblocks;
}

// K = up to MAX_TRIAL_LENGTHS trial branch lengths per sumtable scan; totals in the order
// df[0], ddf[0], df[1], ddf[1], ...  (to out_df / out_ddf, or left at deferred->dst)
int derivatives_impl(pll_partition_t * p, int parent_scaler_index, int child_scaler_index,
                     const double * brlens, unsigned count, const unsigned * params_indices,
                     const double * sumtable, const Engine::Sink * deferred,
                     double * out_df, double * out_ddf)
{
  Engine * e = engine_of(p);
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_scaler_index(e, parent_scaler_index) || !check_scaler_index(e, child_scaler_index))
    return PLL_FAILURE;
  if (!count || count > MAX_TRIAL_LENGTHS)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "1 to %u trial branch lengths per call", MAX_TRIAL_LENGTHS);
    return PLL_FAILURE;
  }
  double * d_sum = sumtable_device(e, sumtable, false);
  if (!d_sum)
  {
    set_error(PLL_ERROR_PARAM_INVALID,
              "pll_compute_likelihood_derivatives: no sumtable was computed for this buffer");
    return PLL_FAILURE;
  }
  if (!sync_model(p) || !ensure_invariant(p)) return PLL_FAILURE;
  if (!need_scaler(e, parent_scaler_index) || !need_scaler(e, child_scaler_index)) return PLL_FAILURE;
  const unsigned nblocks = (e->blocked || e->family == KernelFamily::S4) ? scan_grid(e) : reduce_grid(e);
  const ModelView mv = model_view(e);
  const ParamIdx params = make_params(p, params_indices);
  // lengths per launch: the matrix-core kernel (20 / 61 states) takes four; the others are
  // bounded by their coefficient tables (3 x R x S doubles per length, below 48 KiB of LDS)
  unsigned kmax = MAX_TRIAL_LENGTHS;
  if (e->blocked) kmax = 4;
  else if (e->family != KernelFamily::S4)
  {
    while (kmax > 1 && (size_t)3 * kmax * e->R * e->S * sizeof(double) > 48 * 1024) kmax >>= 1;
    if ((size_t)3 * e->R * e->S * sizeof(double) > 64 * 1024)
    {
      set_error(PLL_ERROR_PARAM_INVALID, "derivatives: %u rates x %u states exceed the LDS tables", e->R, e->S);
      return PLL_FAILURE;
    }
  }
  const bool asc = e->N > e->Nreal;
  if (asc && deferred)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "internal: deferred derivatives of an AB partition go through the blocking form");
    return PLL_FAILURE;
  }
  if (asc)
  {
    // {A, B, C, count} of the constant patterns at every trial length, ahead of the scan on the
    // same stream: when the scan's result flag arrives these are in host memory too
    TrialLengths all;
    for (unsigned i = 0; i < MAX_TRIAL_LENGTHS; ++i) all.t[i] = brlens[std::min(i, count - 1)];
    hipLaunchKernelGGL(k_asc_abc, dim3(1), dim3(256), 0, e->stream, mv, params, all, count, d_sum,
                       scaler_ptr(e, parent_scaler_index), scaler_ptr(e, child_scaler_index),
                       e->Nreal, e->R, e->rows, e->rate_scalers ? 1u : 0u, e->d_asc);
    PLLHIP_TRY(hipGetLastError());
  }
  Engine::Sink base;
  if (deferred) base = *deferred; else { sink_to_host(e); base = e->sink; }
  for (unsigned first = 0; first < count; first += kmax)
  {
    const unsigned nb = std::min(kmax, count - first);
    TrialLengths tl;
    for (unsigned i = 0; i < MAX_TRIAL_LENGTHS; ++i) tl.t[i] = brlens[first + std::min(i, nb - 1)];
    e->sink = base;
    e->sink.dst = base.dst + 2 * first;
    e->sink.nq = 2 * nb;
    if (first + nb < count) e->sink.flag = nullptr;     // the last launch tells the host
    const unsigned * ps = scaler_ptr(e, parent_scaler_index), * cs = scaler_ptr(e, child_scaler_index);
    int rc;
    if (e->family == KernelFamily::S4)
      rc = launch_derivatives_s4(e, mv, params, tl, nb, d_sum, ps, cs, nblocks);
    else if (e->family == KernelFamily::S20)
      rc = launch_derivatives_s20(e, mv, params, tl, nb, d_sum, ps, cs, nblocks);
    else if (e->family == KernelFamily::S61)
      rc = launch_derivatives_s61(e, mv, params, tl, nb, d_sum, ps, cs, nblocks);
    else if (e->family == KernelFamily::S16)
      rc = launch_derivatives_s16(e, mv, params, tl, nb, d_sum, ps, cs, nblocks);
    else
      rc = launch_derivatives_generic(e, mv, params, tl, nb, d_sum, ps, cs, nblocks);
    if (!rc || !finish_launch(e, nblocks, e->blocked ? 8 : 2 * trial_instance(nb))) return PLL_FAILURE;
  }
  e->counters.derivative_calls++;
  e->counters.derivative_points += count;
  if (deferred) return PLL_SUCCESS;
  const volatile unsigned long long * flag =
      reinterpret_cast<const volatile unsigned long long *>(e->h_result) + RESULT_SEQ_SLOT;
  if (!wait_sequence(e->stream, flag, base.seq)) return PLL_FAILURE;
  for (unsigned i = 0; i < count; ++i) { out_df[i] = e->h_result[2 * i]; out_ddf[i] = e->h_result[2 * i + 1]; }
  if (asc)
    for (unsigned i = 0; i < count; ++i)
    {
      // d_f, dd_f are derivatives of -lnL: the correction's derivatives come off
      double d1, d2;
      asc_derivatives(p, e->h_asc + (size_t)i * e->S * 4, &d1, &d2);
      out_df[i] -= d1;
      out_ddf[i] -= d2;
    }
  return PLL_SUCCESS;
}

} // namespace pllhip

extern "C" {

double pll_compute_edge_loglikelihood(pll_partition_t * p,
                                      unsigned int parent_clv_index, int parent_scaler_index,
                                      unsigned int child_clv_index, int child_scaler_index,
                                      unsigned int matrix_index,
                                      const unsigned int * freqs_indices,
                                      double * persite_lnl)
{
  if (is_router(p))
    return router_loglikelihood(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                child_scaler_index, (int)matrix_index, freqs_indices, persite_lnl);
  return loglikelihood_impl(p, parent_clv_index, parent_scaler_index, child_clv_index,
                            child_scaler_index, (int)matrix_index, freqs_indices, persite_lnl, nullptr);
}

double pll_compute_root_loglikelihood(pll_partition_t * p, unsigned int clv_index, int scaler_index,
                                      const unsigned int * freqs_indices, double * persite_lnl)
{
  if (is_router(p))
    return router_loglikelihood(p, clv_index, scaler_index, 0, PLL_SCALE_BUFFER_NONE, -1, freqs_indices, persite_lnl);
  return loglikelihood_impl(p, clv_index, scaler_index, 0, PLL_SCALE_BUFFER_NONE, -1,
                            freqs_indices, persite_lnl, nullptr);
}

int pll_update_sumtable(pll_partition_t * p,
                        unsigned int parent_clv_index, unsigned int child_clv_index,
                        int parent_scaler_index, int child_scaler_index,
                        const unsigned int * params_indices, double * sumtable)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    if (!ensure_eigen(p, params_indices)) return PLL_FAILURE;
    for (pll_partition_t * c : e->shards)
    {
      push_model(p, c);
      if (!pll_update_sumtable(c, parent_clv_index, child_clv_index, parent_scaler_index, child_scaler_index,
                               params_indices, sumtable))
        return PLL_FAILURE;
    }
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_clv_index(e, parent_clv_index, "parent") || !check_clv_index(e, child_clv_index, "child") ||
      !check_scaler_index(e, parent_scaler_index) || !check_scaler_index(e, child_scaler_index))
    return PLL_FAILURE;
  if (!sumtable)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "sumtable key must not be NULL");
    return PLL_FAILURE;
  }
  if (!ensure_eigen(p, params_indices) || !sync_model(p)) return PLL_FAILURE;
  if (e->coded_tips && !ensure_luts(p)) return PLL_FAILURE;
  if (!need_clv(e, parent_clv_index) || !need_clv(e, child_clv_index)) return PLL_FAILURE;
  double * d_sum = sumtable_device(e, sumtable, true);
  if (!d_sum) return PLL_FAILURE;

  const ModelView mv = model_view(e);
  const ParamIdx params = make_params(p, params_indices);
  const NodeRef parent = node_ref(e, parent_clv_index), child = node_ref(e, child_clv_index);
  int rc;
  if (e->family == KernelFamily::S4)
    rc = launch_sumtable_s4(e, mv, params, parent, child, d_sum);
  else if (e->family == KernelFamily::S20)
    rc = launch_sumtable_s20(e, mv, params, parent, child, d_sum);
  else if (e->family == KernelFamily::S61)
    rc = launch_sumtable_s61(e, mv, params, parent, child, d_sum);
  else if (e->family == KernelFamily::S16)
    rc = launch_sumtable_s16(e, mv, params, parent, child, d_sum);
  else
    rc = launch_sumtable_generic(e, mv, params, parent, child, d_sum);
  e->counters.sumtable_calls++;
  return rc;
}

int pll_compute_likelihood_derivatives(pll_partition_t * p,
                                       int parent_scaler_index, int child_scaler_index,
                                       double branch_length,
                                       const unsigned int * params_indices,
                                       const double * sumtable, double * d_f, double * dd_f)
{
  if (is_router(p))
    return router_derivatives(p, parent_scaler_index, child_scaler_index, &branch_length, 1, params_indices,
                              sumtable, d_f, dd_f);
  return derivatives_impl(p, parent_scaler_index, child_scaler_index, &branch_length, 1, params_indices,
                          sumtable, nullptr, d_f, dd_f);
}

} // extern "C"

namespace pllhip {

// one partition's launch of the Newton-Raphson loop, everything resolved
struct NewtonLaunch
{
  Engine * e = nullptr;
  const void * fn = nullptr;
  size_t lds = 0;
  unsigned nblocks = 0;
  double share = 0.0;             // fraction of the chip its workgroups take while they are all resident
  double * d_sum = nullptr;
  const unsigned * ps = nullptr, * cs = nullptr;
  ModelView mv;
  ParamIdx params;
};

// checks and kernel choice of pllhip_newton_branch for one partition.  resident: the loop may keep the sumtable in the
// registers of one-workgroup-per-CU launches (a partition that has the device to itself); otherwise the streaming loop,
// whose workgroups leave room for the other partitions' launches.  Same grid either way: same block totals.
static int newton_prepare(pll_partition_t * p, int parent_scaler_index, int child_scaler_index,
                          const unsigned * params_indices, const double * sumtable, bool resident, NewtonLaunch & L)
{
  Engine * e = engine_of(p);
  static const int enabled = getenv("PLLHIP_DEVICE_NEWTON") ? atoi(getenv("PLLHIP_DEVICE_NEWTON")) : 1;
  const bool family_ok = e->family == KernelFamily::S20 || e->family == KernelFamily::S16 || e->family == KernelFamily::S61 ||
                         (e->family == KernelFamily::S4 && !e->rate_scalers);
  if (!enabled || !e->shards.empty() || !family_ok || e->N > e->Nreal || !e->fused_finish)
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "this partition does not run the Newton-Raphson loop on the device");
    return PLL_FAILURE;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_scaler_index(e, parent_scaler_index) || !check_scaler_index(e, child_scaler_index)) return PLL_FAILURE;
  L.e = e;
  L.d_sum = sumtable_device(e, sumtable, false);
  if (!L.d_sum)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "pllhip_newton_branch: no sumtable was computed for this buffer");
    return PLL_FAILURE;
  }
  if (!sync_model(p) || !ensure_invariant(p)) return PLL_FAILURE;
  if (!need_scaler(e, parent_scaler_index) || !need_scaler(e, child_scaler_index)) return PLL_FAILURE;
  // the scan's own grid (derivatives_impl): the block totals, and with them every bit of the sums, are the same
  L.nblocks = scan_grid(e);
  int capacity = newton_capacity(e, &L.fn, &L.lds);
  if (capacity < 0) return PLL_FAILURE;
  if (!resident && e->newton_resident && e->family != KernelFamily::S4)
  {
    L.fn = e->newton_stream_fn;
    L.lds = e->newton_stream_lds;
    capacity = e->newton_stream_capacity;
  }
  // every workgroup waits for the others inside the launch: all of them have to be on the chip at once
  if ((int)L.nblocks > capacity || capacity == 0)
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "the scan grid (%u workgroups) does not fit the chip at once (%d)",
              L.nblocks, capacity);
    return PLL_FAILURE;
  }
  L.share = (double)L.nblocks / (double)capacity;
  if (!e->d_newton)
  {
    if (!dev_alloc(reinterpret_cast<NewtonControl **>(&e->d_newton), 1, "Newton-Raphson control block")) return PLL_FAILURE;
    PLLHIP_TRY(hipMemsetAsync(e->d_newton, 0, sizeof(NewtonControl), e->stream));     // (`arrived`, `iter`: NewtonParams)
    PLLHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_newton), 128 * sizeof(double), hipHostMallocMapped));
    memset(e->h_newton, 0, 128 * sizeof(double));
    PLLHIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&e->hd_newton), e->h_newton, 0));
  }
  L.mv = model_view(e);
  L.params = make_params(p, params_indices);
  L.ps = scaler_ptr(e, parent_scaler_index);
  L.cs = scaler_ptr(e, child_scaler_index);
  return PLL_SUCCESS;
}

// the launch: totals of the partition's scans go to its OWN control block's `tot`, everything else (iterate, bracket,
// status, trail, the partitions' meeting point) lives in `ctl`, results in host_out / host_flag
static int newton_launch(const NewtonLaunch & L, const NewtonParams & np_, NewtonControl * ctl, double * host_out,
                         unsigned long long * host_flag, unsigned long long seq)
{
  Engine * e = L.e;
  NewtonParams np = np_;
  ReduceOut ro;
  ro.block_out = e->d_partials;
  ro.counter = e->d_counter;
  ro.dst = static_cast<NewtonControl *>(e->d_newton)->tot;
  ro.flag = nullptr;
  ro.seq = 0;
  ro.fused = 1;
  ro.nq = 2;
  unsigned rs = e->rate_scalers ? 1u : 0u;
  if (e->family == KernelFamily::S4)
  {
    void * args4[] = {(void *)&L.mv, (void *)&L.params, (void *)&np, (void *)&L.d_sum, (void *)&L.ps, (void *)&L.cs,
                      (void *)&e->d_weights, (void *)&e->d_invariant, (void *)&e->N, (void *)&e->R,
                      (void *)&ro, (void *)&ctl, (void *)&host_out, (void *)&host_flag, (void *)&seq};
    PLLHIP_TRY(hipLaunchKernel(L.fn, dim3(L.nblocks), dim3(256), args4, 0, e->stream));
  }
  else
  {
    void * args[] = {(void *)&L.mv, (void *)&L.params, (void *)&np, (void *)&L.d_sum, (void *)&L.ps, (void *)&L.cs,
                     (void *)&e->d_weights, (void *)&e->d_invariant, (void *)&e->N, (void *)&e->nblk, (void *)&e->R,
                     (void *)&ro, (void *)&rs, (void *)&ctl, (void *)&host_out, (void *)&host_flag, (void *)&seq};
    PLLHIP_TRY(hipLaunchKernel(L.fn, dim3(L.nblocks), dim3(256), args, L.lds, e->stream));
  }
  e->counters.derivative_calls++;
  return PLL_SUCCESS;
}

static NewtonParams newton_params(double start, double bl_min, double bl_max, double tolerance, unsigned max_newton,
                                  unsigned nblocks_first)
{
  NewtonParams np;
  np.bl_min = bl_min; np.bl_max = bl_max; np.tolerance = tolerance;
  np.dxmax = bl_max / max_newton;
  np.max_newton = max_newton;
  np.x0 = std::max(std::min(start, bl_max), bl_min);
  // PLLHIP_NEWTON_SPIN_LIMIT: polls before a waiting workgroup gives up; PLLHIP_FAULT=newton_stall: workgroup 0 of
  // the next PLLHIP_FAULT_COUNT (default 1) launches never arrives (tests of the PLLHIP_ERROR_NEWTON_STUCK path)
  static const unsigned env_spin = getenv("PLLHIP_NEWTON_SPIN_LIMIT") ? (unsigned)strtoul(getenv("PLLHIP_NEWTON_SPIN_LIMIT"), nullptr, 10) : 0u;
  static int stall_left = (getenv("PLLHIP_FAULT") && !strcmp(getenv("PLLHIP_FAULT"), "newton_stall"))
                              ? (getenv("PLLHIP_FAULT_COUNT") ? atoi(getenv("PLLHIP_FAULT_COUNT")) : 1) : 0;
  np.spin_limit = env_spin ? env_spin : NEWTON_SPIN_LIMIT;
  np.stall_block = ~0u;
  if (stall_left > 0 && nblocks_first > 1) { --stall_left; np.stall_block = 0u; }
  np.part = 0;
  np.nparts = 1;
  np.xscale = 1.0;
  np.debug = getenv("PLLHIP_NEWTON_DEBUG") ? 1u : 0u;
  np.iter_base = 0;
  return np;
}

// wait for the loop that `lead` hosts and translate how it ended; `all`: every engine that took part
static int newton_finish(Engine * lead, const std::vector<Engine *> & all, unsigned long long seq,
                         double * length, unsigned int * iterations, double * trail)
{
  const volatile unsigned long long * flag = reinterpret_cast<const volatile unsigned long long *>(lead->h_newton + 112);
  if (!wait_sequence(lead->stream, flag, seq)) return PLL_FAILURE;
  const unsigned its = (unsigned)lead->h_newton[1], status = (unsigned)lead->h_newton[2];
  if (status == NEWTON_STUCK)
  {
    // the grids drain by themselves (every wait is bounded); the tickets of the unfinished reductions go back to
    // zero before anything else uses them
    for (Engine * e : all)
    {
      PLLHIP_TRY(hipSetDevice(e->device));
      PLLHIP_TRY(hipStreamSynchronize(e->stream));
      if (getenv("PLLHIP_NEWTON_DEBUG"))
      {
        // where the loop stood: the meeting point and every instance's tickets
        NewtonControl c;
        std::vector<unsigned> t(REDUCE_COUNTER_WORDS);
        (void)hipMemcpy(&c, lead->d_newton, sizeof(c), hipMemcpyDeviceToHost);
        (void)hipMemcpy(t.data(), e->d_counter, t.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        fprintf(stderr, "newton stuck: instance %p iter %u status %u arrived %u x %g | tickets", (void *)e, c.iter, c.status, c.arrived, c.x);
        for (unsigned k = 0; k <= REDUCE_SHARDS; ++k) fprintf(stderr, " %u", t[k * REDUCE_SHARD_STRIDE]);
        fprintf(stderr, " | ptot");
        for (unsigned k = 0; k < all.size(); ++k) fprintf(stderr, " (%g %g)", c.ptot[k][0], c.ptot[k][1]);
        fprintf(stderr, " | waits entered / left / arrived-after-scan per instance:");
        for (unsigned k = 0; k < all.size(); ++k) fprintf(stderr, " %u/%u/%u", c.dbg_enter[k], c.dbg_leave[k], c.dbg_arrive[k]);
        fprintf(stderr, "\n");
      }
      PLLHIP_TRY(hipMemsetAsync(e->d_counter, 0, REDUCE_COUNTER_WORDS * sizeof(unsigned), e->stream));
      PLLHIP_TRY(hipStreamSynchronize(e->stream));
    }
    // ... and the meeting point of the partitions (`arrived`) with them: the next loop does not initialise it
    PLLHIP_TRY(hipSetDevice(lead->device));
    PLLHIP_TRY(hipMemsetAsync(lead->d_newton, 0, sizeof(NewtonControl), lead->stream));
    PLLHIP_TRY(hipStreamSynchronize(lead->stream));
    set_error(PLLHIP_ERROR_NEWTON_STUCK, "the device-resident Newton-Raphson loop did not get all its workgroups onto the "
              "chip at once (the device is shared with other work)");
    return PLL_FAILURE;
  }
  for (Engine * e : all) e->counters.derivative_points += its;
  if (length) *length = lead->h_newton[0];
  if (iterations) *iterations = its;
  if (trail) for (unsigned i = 0; i < its && i < NEWTON_TRAIL_MAX; ++i) trail[i] = lead->h_newton[NEWTON_TRAIL_SLOT + i];
  switch (status)
  {
    case NEWTON_CONVERGED: return PLL_SUCCESS;
    case NEWTON_LIMIT: set_error(PLLHIP_ERROR_NEWTON_LIMIT, "Exceeded maximum number of iterations"); return PLL_FAILURE;
    case NEWTON_NONFINITE: set_error(PLLHIP_ERROR_NEWTON_DERIVATIVES, "Wrong likelihood derivatives"); return PLL_FAILURE;
    default: set_error(PLL_ERROR_HIP_RUNTIME, "the device-resident Newton-Raphson loop did not complete (status %u)", status);
             return PLL_FAILURE;
  }
}

} // namespace pllhip

extern "C" {

int pllhip_newton_branch(pll_partition_t * p, int parent_scaler_index, int child_scaler_index,
                         const unsigned int * params_indices, const double * sumtable,
                         double start, double bl_min, double bl_max, double tolerance, unsigned int max_newton,
                         double * length, unsigned int * iterations, double * trail)
{
  if (!max_newton)
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "this partition does not run the Newton-Raphson loop on the device");
    return PLL_FAILURE;
  }
  NewtonLaunch L;
  if (!newton_prepare(p, parent_scaler_index, child_scaler_index, params_indices, sumtable, true, L)) return PLL_FAILURE;
  Engine * e = L.e;
  const NewtonParams np = newton_params(start, bl_min, bl_max, tolerance, max_newton, L.nblocks);
  NewtonControl init;
  memset(&init, 0, sizeof(init));
  init.x = np.x0; init.xl = bl_min; init.xh = bl_max; init.iter = 0; init.status = NEWTON_RUNNING;
  NewtonControl * ctl = static_cast<NewtonControl *>(e->d_newton);
  // (the control block is initialised from the kernel arguments -- NewtonParams::iter_base --; with PLLHIP_NEWTON_DEBUG
  // the counters of the dump are reset by a copy: pageable source, staged by the runtime before the call returns)
  const unsigned long long seq = ++e->newton_seq;
  NewtonParams np1 = np;
  np1.iter_base = (unsigned)(seq << 8);
  if (np1.debug) PLLHIP_TRY(hipMemcpyAsync(ctl, &init, sizeof(init), hipMemcpyHostToDevice, e->stream));
  if (!newton_launch(L, np1, ctl, e->hd_newton, reinterpret_cast<unsigned long long *>(e->hd_newton + 112), seq)) return PLL_FAILURE;
  return newton_finish(e, std::vector<Engine *>(1, e), seq, length, iterations, trail);
}

// The loop over several partitions runs one launch per partition, each on its partition's stream, and the launches wait
// for one another on the device: they have to be there TOGETHER.  Two streams that share a hardware queue run one after
// the other -- with five streams on the runtime's four queues the second instance started when the first had given up
// (PLLHIP_NEWTON_DEBUG=1 shows it) -- so the loop is offered only to a process that asked the runtime for enough queues
// before its first HIP call: GPU_MAX_HW_QUEUES >= 8 in the environment (the library does not set it: see
// pllhip_runtime_defaults for what it costs).  Elsewhere pllhip_newton_branch_multi answers
// PLLHIP_ERROR_NEWTON_UNSUPPORTED and the callers iterate from the host.
static bool newton_multi_enabled()
{
  static const bool on = []() { const char * q = getenv("GPU_MAX_HW_QUEUES"); return q && atoi(q) >= 8; }();
  return on;
}

// The loop over several partitions as ONE launch (k_newton_multi, kernels_newton_s4.hpp): every partition a run of the
// launch's workgroups.  -1: not for these partitions (a family without a place in the kernel, a 4-state partition whose
// table does not fit the registers, grids that do not fit the chip together): the caller tries the other forms.
static int newton_multi_one_launch(pll_partition_t * const * partitions, unsigned count, int parent_scaler_index,
                                   int child_scaler_index, const unsigned int * const * params_indices,
                                   const double * const * sumtables, const double * length_scalers,
                                   double start, double bl_min, double bl_max, double tolerance, unsigned max_newton,
                                   double * length, unsigned * iterations, double * trail)
{
  static const int enabled = getenv("PLLHIP_NEWTON_ONE_LAUNCH") ? atoi(getenv("PLLHIP_NEWTON_ONE_LAUNCH")) : 1;
  if (!enabled) return -1;
  for (unsigned k = 0; k < count; ++k)
  {
    const Engine * e = engine_of(partitions[k]);
    if (e->family != KernelFamily::S4 && e->family != KernelFamily::S20 && e->family != KernelFamily::S61 &&
        e->family != KernelFamily::S16) return -1;
  }
  std::vector<NewtonLaunch> L(count);
  NewtonMultiArgs args;
  memset(&args, 0, sizeof(args));
  unsigned total = 0;
  size_t lds = 0;
  const int saved_errno = pll_errno;
  for (unsigned k = 0; k < count; ++k)
  {
    // (the streaming form of the matrix-core families: the sumtable of a slice fits the registers of a launch of its
    // own, not of a shared one)
    if (!newton_prepare(partitions[k], parent_scaler_index, child_scaler_index, params_indices[k], sumtables[k], false, L[k]))
    {
      if (pll_errno != PLLHIP_ERROR_NEWTON_UNSUPPORTED) return PLL_FAILURE;
      pll_errno = saved_errno;
      return -1;
    }
    Engine * e = L[k].e;
    NewtonMultiPart & P = args.part[k];
    P.mv = L[k].mv; P.params = L[k].params; P.sumtable = L[k].d_sum; P.ps = L[k].ps; P.cs = L[k].cs;
    P.weights = e->d_weights; P.invariant = e->d_invariant;
    P.ro.block_out = e->d_partials; P.ro.counter = e->d_counter; P.ro.dst = static_cast<NewtonControl *>(e->d_newton)->tot;
    P.ro.flag = nullptr; P.ro.seq = 0; P.ro.fused = 1; P.ro.nq = 2;
    P.xscale = length_scalers ? length_scalers[k] : 1.0;
    P.N = e->N; P.nblk = e->nblk; P.R = e->R; P.rate_scalers = e->rate_scalers ? 1u : 0u;
    P.kind = e->family == KernelFamily::S4 ? NEWTON_KIND_S4 : e->family == KernelFamily::S20 ? NEWTON_KIND_S20
           : e->family == KernelFamily::S16 ? NEWTON_KIND_S16 + s16_ks(e) - 1
           : e->S == S61_S ? NEWTON_KIND_S61 : NEWTON_KIND_S61_RT;
    P.first_block = total; P.nblocks = L[k].nblocks;
    total += L[k].nblocks;
    if (e->family != KernelFamily::S4)
      lds = std::max(lds, sizeof(double) * e->R * (e->family == KernelFamily::S20 ? 5u : e->family == KernelFamily::S16 ? s16_ks(e) : S61_KS) * 64);
  }
  Engine * lead = L[0].e;
  PLLHIP_TRY(hipSetDevice(lead->device));
  if (lds > 64 * 1024) return -1;
  // every workgroup waits for the others inside the launch: all of them on the chip at once
  bool any16 = false;
  for (unsigned k = 0; k < count; ++k) any16 = any16 || L[k].e->family == KernelFamily::S16;
  const void * fn = any16 ? reinterpret_cast<const void *>(k_newton_multi<true>) : reinterpret_cast<const void *>(k_newton_multi<false>);
  static int per_cu_cache[2][2] = {{-1, -1}, {-1, -1}};    // (kernel; without / with dynamic LDS beyond 16 KiB)
  int & per_cu = per_cu_cache[any16 ? 1 : 0][lds > 16 * 1024 ? 1 : 0];
  if (per_cu < 0 &&
      !hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256,
                                                          lds > 16 * 1024 ? 64 * 1024 : 16 * 1024), "hipOccupancyMaxActiveBlocksPerMultiprocessor"))
    return PLL_FAILURE;
  // (one launch: the dispatcher fills the chip with its workgroups alone, so all of what the chip holds may be asked for
  // -- as in the loop of a single partition)
  if (per_cu <= 0 || total > (unsigned)per_cu * lead->cu_count) return -1;
  args.np = newton_params(start, bl_min, bl_max, tolerance, max_newton, L[0].nblocks);
  args.np.nparts = count;
  args.nparts = count;
  NewtonControl init;
  memset(&init, 0, sizeof(init));
  init.x = args.np.x0; init.xl = bl_min; init.xh = bl_max; init.iter = 0; init.status = NEWTON_RUNNING;
  for (unsigned k = 0; k < count; ++k) init.pscale[k] = args.part[k].xscale;
  NewtonControl * ctl = static_cast<NewtonControl *>(lead->d_newton);
  // the launch runs on the first partition's stream: after what the others have issued (their sumtables) ...
  for (unsigned k = 1; k < count; ++k)
  {
    Engine * e = L[k].e;
    if (e->stream == lead->stream) continue;
    if (!e->newton_ready) PLLHIP_TRY(hipEventCreateWithFlags(&e->newton_ready, hipEventDisableTiming));
    PLLHIP_TRY(hipEventRecord(e->newton_ready, e->stream));
    PLLHIP_TRY(hipStreamWaitEvent(lead->stream, e->newton_ready, 0));
    // (... and after the instance of an earlier loop of the other form, should there have been one)
    if (e->newton_done) PLLHIP_TRY(hipStreamWaitEvent(lead->stream, e->newton_done, 0));
  }
  const unsigned long long seq = ++lead->newton_seq;
  args.np.iter_base = (unsigned)(seq << 8);
  if (args.np.debug) PLLHIP_TRY(hipMemcpyAsync(ctl, &init, sizeof(init), hipMemcpyHostToDevice, lead->stream));
  double * host_out = lead->hd_newton;
  unsigned long long * host_flag = reinterpret_cast<unsigned long long *>(lead->hd_newton + 112);
  void * kargs[] = {(void *)&args, (void *)&ctl, (void *)&host_out, (void *)&host_flag, (void *)&seq};
  PLLHIP_TRY(hipLaunchKernel(fn, dim3(total), dim3(256), kargs, lds, lead->stream));
  std::vector<Engine *> all;
  for (unsigned k = 0; k < count; ++k) { all.push_back(L[k].e); L[k].e->counters.derivative_calls++; }
  // ... and what the others issue next comes after it
  if (!lead->newton_done) PLLHIP_TRY(hipEventCreateWithFlags(&lead->newton_done, hipEventDisableTiming));
  PLLHIP_TRY(hipEventRecord(lead->newton_done, lead->stream));
  for (unsigned k = 1; k < count; ++k)
    if (L[k].e->stream != lead->stream) PLLHIP_TRY(hipStreamWaitEvent(L[k].e->stream, lead->newton_done, 0));
  return newton_finish(lead, all, seq, length, iterations, trail);
}

// Several partitions under ONE branch length (linked lengths, or scaled ones: partition p sees s_p x): the loop of
// pllhip_newton_branch with the sum over the partitions inside it.  Every partition launches its own instance of the
// loop on its own stream -- its family's kernel on its own scan grid, so its totals are the ones its blocking derivative
// call returns, bit for bit --; the instances meet in the first partition's control block after every scan (the
// partition that arrives last adds the totals in partition order and applies the step rule) and all wait there for the
// next iterate.  All launches have to be resident together: their shares of the chip must add up to less than one
// (with the register-resident form where that fits, else the streaming form), otherwise PLLHIP_ERROR_NEWTON_UNSUPPORTED.
int pllhip_newton_branch_multi(pll_partition_t * const * partitions, unsigned int count,
                               int parent_scaler_index, int child_scaler_index,
                               const unsigned int * const * params_indices, const double * const * sumtables,
                               const double * length_scalers,
                               double start, double bl_min, double bl_max, double tolerance, unsigned int max_newton,
                               double * length, unsigned int * iterations, double * trail)
{
  if (count == 1 && (!length_scalers || length_scalers[0] == 1.0))
    return pllhip_newton_branch(partitions[0], parent_scaler_index, child_scaler_index, params_indices[0], sumtables[0],
                                start, bl_min, bl_max, tolerance, max_newton, length, iterations, trail);
  if (!count || count > NEWTON_MAX_PARTS || !max_newton)
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "1 to %u partitions per device-resident Newton-Raphson loop", NEWTON_MAX_PARTS);
    return PLL_FAILURE;
  }
  for (unsigned k = 0; k < count; ++k)
    if (!partitions[k] || engine_of(partitions[k])->device != engine_of(partitions[0])->device)
    {
      set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "the partitions of a device-resident Newton-Raphson loop live on one device");
      return PLL_FAILURE;
    }
  // all partitions in ONE launch where their families have a place in it (4, 20, 33 .. 64 states): no hardware queue
  // per partition needed
  {
    const int rc = newton_multi_one_launch(partitions, count, parent_scaler_index, child_scaler_index, params_indices, sumtables,
                                           length_scalers, start, bl_min, bl_max, tolerance, max_newton, length, iterations, trail);
    if (rc >= 0) return rc;
  }
  if (!newton_multi_enabled())
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "the Newton-Raphson loop over these partitions runs as one launch per partition and "
              "needs a hardware queue per partition stream: GPU_MAX_HW_QUEUES >= 8 in the environment before the first HIP call");
    return PLL_FAILURE;
  }
  // (a launch's share = its workgroups / the workgroups of its kernel the chip holds: an upper bound of the resources
  // it takes, so shares that add up to one fit; the margin is for the order in which the dispatcher fills the CUs.
  // A wrong guess costs one bounded wait: PLLHIP_ERROR_NEWTON_STUCK, and the caller's own loop from then on)
  static const double max_share = getenv("PLLHIP_NEWTON_MAX_SHARE") ? atof(getenv("PLLHIP_NEWTON_MAX_SHARE")) : 0.9;
  std::vector<NewtonLaunch> L(count);
  bool fits = false;
  for (int resident = 1; resident >= 0 && !fits; --resident)
  {
    double share = 0.0;
    for (unsigned k = 0; k < count; ++k)
    {
      L[k] = NewtonLaunch();
      if (!newton_prepare(partitions[k], parent_scaler_index, child_scaler_index, params_indices[k], sumtables[k],
                          resident != 0, L[k]))
        return PLL_FAILURE;
      share += L[k].share;
    }
    fits = share <= max_share + 1e-9;
  }
  if (!fits)
  {
    set_error(PLLHIP_ERROR_NEWTON_UNSUPPORTED, "the scan grids of the %u partitions do not fit the chip together", count);
    return PLL_FAILURE;
  }
  Engine * lead = L[0].e;
  NewtonParams np = newton_params(start, bl_min, bl_max, tolerance, max_newton, L[0].nblocks);
  NewtonControl init;
  memset(&init, 0, sizeof(init));
  init.x = np.x0; init.xl = bl_min; init.xh = bl_max; init.iter = 0; init.status = NEWTON_RUNNING;
  for (unsigned k = 0; k < count; ++k) init.pscale[k] = length_scalers ? length_scalers[k] : 1.0;
  PLLHIP_TRY(hipSetDevice(lead->device));
  // (the instances are different kernels on different hardware queues -- pllhip_runtime_defaults asks the runtime for
  // enough of them; they meet in the first partition's control block)
  NewtonControl * ctl = static_cast<NewtonControl *>(lead->d_newton);
  // (the instances of the previous loop of these partitions have left the device: they read the block to the end)
  for (unsigned k = 1; k < count; ++k)
    if (L[k].e->newton_done) PLLHIP_TRY(hipStreamWaitEvent(lead->stream, L[k].e->newton_done, 0));
  const unsigned long long seq = ++lead->newton_seq;
  np.iter_base = (unsigned)(seq << 8);
  if (np.debug) PLLHIP_TRY(hipMemcpyAsync(ctl, &init, sizeof(init), hipMemcpyHostToDevice, lead->stream));
  // the other partitions' launches use the control block: after the instances of the previous loop have left it
  if (!lead->newton_ready) PLLHIP_TRY(hipEventCreateWithFlags(&lead->newton_ready, hipEventDisableTiming));
  PLLHIP_TRY(hipEventRecord(lead->newton_ready, lead->stream));
  unsigned long long * host_flag = reinterpret_cast<unsigned long long *>(lead->hd_newton + 112);
  std::vector<Engine *> all;
  np.nparts = count;
  // (an instance that shares a hardware queue with another one never meets it: a shorter bound than the single loop's)
  static const bool env_spin_set = getenv("PLLHIP_NEWTON_SPIN_LIMIT") != nullptr;
  if (!env_spin_set) np.spin_limit = NEWTON_SPIN_LIMIT >> 4;
  for (unsigned k = 0; k < count; ++k)
  {
    np.part = k;
    np.xscale = init.pscale[k];
    if (k) { np.stall_block = ~0u; PLLHIP_TRY(hipStreamWaitEvent(L[k].e->stream, lead->newton_ready, 0)); }
    all.push_back(L[k].e);
    if (!newton_launch(L[k], np, ctl, lead->hd_newton, host_flag, seq))
    {
      // (the instances already launched give up after their bounded wait)
      (void)newton_finish(lead, all, seq, nullptr, nullptr, nullptr);
      return PLL_FAILURE;
    }
    if (k)
    {
      if (!L[k].e->newton_done) PLLHIP_TRY(hipEventCreateWithFlags(&L[k].e->newton_done, hipEventDisableTiming));
      PLLHIP_TRY(hipEventRecord(L[k].e->newton_done, L[k].e->stream));
    }
  }
  return newton_finish(lead, all, seq, length, iterations, trail);
}

unsigned int pllhip_free_trial_lengths(const pll_partition_t * p)
{
  return exec_engine(p)->blocked ? 4u : 1u;
}

int pllhip_set_sharding(unsigned int count, const int * devices)
{
  g_shard_devices.clear();
  g_shard_devices_set = true;
  if (count <= 1) return PLL_SUCCESS;
  const int n = pllhip_device_count();
  if (n < 1)
  {
    set_error(PLL_ERROR_HIP_NODEVICE, "No HIP device available");
    return PLL_FAILURE;
  }
  for (unsigned k = 0; k < count; ++k) g_shard_devices.push_back(devices ? devices[k] : (int)(k % (unsigned)n));
  return PLL_SUCCESS;
}

unsigned int pllhip_shard_count(const pll_partition_t * p)
{
  return is_router(p) ? (unsigned)engine_of(p)->shards.size() : 1u;
}

int pllhip_compute_likelihood_derivatives_multi(pll_partition_t * p,
                                                int parent_scaler_index, int child_scaler_index,
                                                const double * branch_lengths, unsigned int count,
                                                const unsigned int * params_indices,
                                                const double * sumtable, double * d_f, double * dd_f)
{
  if (is_router(p))
  {
    if (!count || count > MAX_TRIAL_LENGTHS)
    {
      set_error(PLL_ERROR_PARAM_INVALID, "1 to %u trial branch lengths per call", MAX_TRIAL_LENGTHS);
      return PLL_FAILURE;
    }
    return router_derivatives(p, parent_scaler_index, child_scaler_index, branch_lengths, count, params_indices,
                              sumtable, d_f, dd_f);
  }
  return derivatives_impl(p, parent_scaler_index, child_scaler_index, branch_lengths, count, params_indices,
                          sumtable, nullptr, d_f, dd_f);
}

int pll_update_invariant_sites(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    if (!p->invariant) p->invariant = static_cast<int *>(malloc(sizeof(int) * (p->sites ? p->sites : 1)));
    if (!p->invariant) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate invariant sites array"); return PLL_FAILURE; }
    for (size_t k = 0; k < e->shards.size(); ++k)
    {
      if (!pll_update_invariant_sites(e->shards[k])) return PLL_FAILURE;
      memcpy(p->invariant + e->shard_first[k], e->shards[k]->invariant, sizeof(int) * e->shards[k]->sites);
    }
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!upload_tipmap(p)) return PLL_FAILURE;
  if (!p->invariant)
    p->invariant = static_cast<int *>(malloc(sizeof(int) * (e->N ? e->N : 1)));
  if (!p->invariant)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate invariant sites array");
    return PLL_FAILURE;
  }
  if (!e->d_invariant)
    if (!dev_alloc(&e->d_invariant, (size_t)e->N, "invariant sites")) return PLL_FAILURE;
  // device tables of tip pointers
  std::vector<const double *> h_clv(e->tips ? e->tips : 1, nullptr);
  std::vector<const uint8_t *> h_codes(e->tips ? e->tips : 1, nullptr);
  for (unsigned t = 0; t < e->tips; ++t) { h_clv[t] = e->d_clv[t]; h_codes[t] = e->coded_tips ? e->d_codes[t] : nullptr; }
  const double ** d_tc = nullptr;
  const uint8_t ** d_tk = nullptr;
  if (!dev_alloc(&d_tc, h_clv.size(), "tip table") || !dev_alloc(&d_tk, h_codes.size(), "tip table"))
    return PLL_FAILURE;
  PLLHIP_TRY(hipMemcpyAsync(d_tc, h_clv.data(), sizeof(void *) * h_clv.size(), hipMemcpyHostToDevice, e->stream));
  PLLHIP_TRY(hipMemcpyAsync(d_tk, h_codes.data(), sizeof(void *) * h_codes.size(), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_invariant, dim3(reduce_grid(e)), dim3(256), 0, e->stream,
                     d_tc, d_tk, e->d_tipmap, e->tips, e->N, e->R, e->S, e->Sp, e->rows, e->d_invariant);
  PLLHIP_TRY(hipGetLastError());
  if (e->N)
    PLLHIP_TRY(hipMemcpyAsync(p->invariant, e->d_invariant, sizeof(int) * e->N, hipMemcpyDeviceToHost, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(d_tc);
  (void)hipFree(d_tk);
  e->invariant_uploaded = true;
  return PLL_SUCCESS;
}

int pll_compute_node_ancestral(pll_partition_t * p, unsigned int node_clv_index, int node_scaler_index,
                               unsigned int other_clv_index, int other_scaler_index,
                               unsigned int matrix_index, const unsigned int * freqs_indices,
                               double * ancestral)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
    {
      push_model(p, e->shards[k]);
      if (!pll_compute_node_ancestral(e->shards[k], node_clv_index, node_scaler_index, other_clv_index,
                                      other_scaler_index, matrix_index, freqs_indices,
                                      ancestral + (size_t)e->shard_first[k] * e->S))
        return PLL_FAILURE;
    }
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_clv_index(e, node_clv_index, "node") || !check_clv_index(e, other_clv_index, "other") ||
      !check_scaler_index(e, node_scaler_index) || !check_scaler_index(e, other_scaler_index))
    return PLL_FAILURE;
  if (matrix_index >= e->nmat)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "matrix index out of range");
    return PLL_FAILURE;
  }
  if (!sync_model(p) || !flush_pmatrices(p) || !ensure_luts(p)) return PLL_FAILURE;
  if (!e->N) return PLL_SUCCESS;
  if (!need_clv(e, node_clv_index) || !need_clv(e, other_clv_index)) return PLL_FAILURE;
  // the result is normalised per site, so scaler counts cancel
  double * d_out = nullptr;
  if (!dev_alloc(&d_out, (size_t)e->N * e->S, "ancestral states")) return PLL_FAILURE;
  const unsigned gx = (unsigned)std::min<unsigned long long>(((unsigned long long)e->N + 255) / 256, 4096);
  hipLaunchKernelGGL(k_node_ancestral, dim3(gx), dim3(256), 0, e->stream,
                     model_view(e), make_params(p, freqs_indices), node_ref(e, node_clv_index),
                     node_ref(e, other_clv_index),
                     e->d_pmat + (size_t)matrix_index * e->R * e->S * e->Sp, e->d_tipmap, e->rows,
                     e->N, e->R, d_out);
  PLLHIP_TRY(hipGetLastError());
  PLLHIP_TRY(hipMemcpyAsync(ancestral, d_out, sizeof(double) * e->N * e->S, hipMemcpyDeviceToHost, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(d_out);
  return PLL_SUCCESS;
}

// ===========================================================================
// pllhip_* extension API
// ===========================================================================

int pllhip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pllhip_set_device(int device)
{
  int n = pllhip_device_count();
  if (device < 0 || device >= n)
  {
    set_error(PLL_ERROR_HIP_NODEVICE, "HIP device %d requested, %d visible", device, n);
    return PLL_FAILURE;
  }
  g_device = device;
  return PLL_SUCCESS;
}

int pllhip_get_device(void) { return current_device(); }

int pllhip_device_arch(int device, char * out, size_t out_len)
{
  hipDeviceProp_t prop;
  if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return PLL_FAILURE;
  snprintf(out, out_len, "%s", prop.gcnArchName);
  return PLL_SUCCESS;
}

int pllhip_synchronize(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (pll_partition_t * c : e->shards) if (!pllhip_synchronize(c)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  return PLL_SUCCESS;
}

void * pllhip_stream(pll_partition_t * p) { return exec_engine(p)->stream; }

int pllhip_get_clv(pll_partition_t * p, unsigned int clv_index, double * out)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    // API layout is site-major: a shard's vector is a contiguous piece of the whole
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!pllhip_get_clv(e->shards[k], clv_index, out + (size_t)e->shard_first[k] * e->R * e->Sp)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_clv_index(e, clv_index, "requested")) return PLL_FAILURE;
  const size_t len = (size_t)e->N * e->R * e->Sp;
  if (!len) return PLL_SUCCESS;
  if (clv_index < e->tips && e->coded_tips)
  {
    double * tmp = nullptr;
    if (!upload_tipmap(p) || !dev_alloc(&tmp, len, "expanded tip")) return PLL_FAILURE;
    hipLaunchKernelGGL(k_expand_codes, dim3(reduce_grid(e)), dim3(256), 0, e->stream,
                       e->d_codes[clv_index], e->d_tipmap, e->N, e->R, e->S, e->Sp, tmp);
    PLLHIP_TRY(hipGetLastError());
    PLLHIP_TRY(hipMemcpyAsync(out, tmp, len * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(tmp);
    return PLL_SUCCESS;
  }
  if (!need_clv(e, clv_index)) return PLL_FAILURE;
  return fetch_clv(e, e->d_clv[clv_index], out);
}

int pllhip_set_clv(pll_partition_t * p, unsigned int clv_index, const double * clv)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!pllhip_set_clv(e->shards[k], clv_index, clv + (size_t)e->shard_first[k] * e->R * e->Sp)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (!check_clv_index(e, clv_index, "target")) return PLL_FAILURE;
  if (clv_index < e->tips && e->coded_tips)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "CLV %u is a coded tip", clv_index);
    return PLL_FAILURE;
  }
  if (!transient_flush_all(e)) return PLL_FAILURE;            // (vectors that were not stored may read this one)
  if (e->shadow_codes && clv_index < e->tips) e->tip_has_codes[clv_index] = 0;
  if (!e->cherries.empty())
  {
    e->cherries[clv_index].valid = false;
    if (clv_index < e->tips) { e->tip_version[clv_index] = ++e->class_clock; e->plan.key.clear(); }
  }
  return store_clv(e, e->d_clv[clv_index], clv);
}

int pllhip_set_transient(pll_partition_t * p, int enable)
{
  Engine * e = engine_of(p);
  for (pll_partition_t * c : e->shards) (void)pllhip_set_transient(c, enable);
  e->transient_mode = enable != 0;
  return PLL_SUCCESS;
}

int pllhip_discard_transient(pll_partition_t * p)
{
  Engine * e = engine_of(p);
  for (pll_partition_t * c : e->shards) (void)pllhip_discard_transient(c);
  for (unsigned n = 0; n < e->nodes && e->ntransient; ++n) transient_drop(e, n, true);
  return PLL_SUCCESS;
}

int pllhip_transient_stats(const pll_partition_t * p, pllhip_transient_stats_t * out)
{
  *out = exec_engine(p)->transient_stats;
  return PLL_SUCCESS;
}

int pllhip_repeat_stats(const pll_partition_t * p, pllhip_repeat_stats_t * out)
{
  *out = exec_engine(p)->repeat_stats;
  return PLL_SUCCESS;
}

int pllhip_get_scaler(pll_partition_t * p, unsigned int idx, unsigned int * out)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    const size_t per = e->rate_scalers ? e->R : 1;
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!pllhip_get_scaler(e->shards[k], idx, out + (size_t)e->shard_first[k] * per)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (idx >= e->nscalers) { set_error(PLL_ERROR_PARAM_INVALID, "scaler index out of range"); return PLL_FAILURE; }
  if (!e->N) return PLL_SUCCESS;
  if (!need_scaler(e, (int)idx)) return PLL_FAILURE;
  PLLHIP_TRY(hipMemcpyAsync(out, e->d_scalers + (size_t)idx * e->sc_len,
                            sizeof(unsigned) * e->N * (e->rate_scalers ? e->R : 1),
                            hipMemcpyDeviceToHost, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  return PLL_SUCCESS;
}

int pllhip_set_scaler(pll_partition_t * p, unsigned int idx, const unsigned int * in)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    const size_t per = e->rate_scalers ? e->R : 1;
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!pllhip_set_scaler(e->shards[k], idx, in + (size_t)e->shard_first[k] * per)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  if (idx >= e->nscalers) { set_error(PLL_ERROR_PARAM_INVALID, "scaler index out of range"); return PLL_FAILURE; }
  if (!e->N) return PLL_SUCCESS;
  if (!transient_flush_all(e)) return PLL_FAILURE;
  if (idx < e->scaler_lazy.size()) e->scaler_lazy[idx] = -1;         // the caller's counts replace whatever stood for the buffer
  PLLHIP_TRY(hipMemcpyAsync(e->d_scalers + (size_t)idx * e->sc_len, in,
                            sizeof(unsigned) * e->N * (e->rate_scalers ? e->R : 1),
                            hipMemcpyHostToDevice, e->stream));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  return PLL_SUCCESS;
}

int pllhip_get_sumtable(pll_partition_t * p, const double * key, double * out)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    for (size_t k = 0; k < e->shards.size(); ++k)
      if (!pllhip_get_sumtable(e->shards[k], key, out + (size_t)e->shard_first[k] * e->R * e->Sp)) return PLL_FAILURE;
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  double * d_sum = sumtable_device(e, key, false);
  if (!d_sum) { set_error(PLL_ERROR_PARAM_INVALID, "unknown sumtable key"); return PLL_FAILURE; }
  return fetch_clv(e, d_sum, out);
}

int pllhip_sync_to_host(pll_partition_t * p, unsigned int what)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty() && (what & PLLHIP_SYNC_PMATRIX) && e->nmat)
  {
    // every shard holds the same matrices: the first one's mirror is the partition's
    pll_partition_t * c = e->shards[0];
    if (!pllhip_sync_to_host(c, PLLHIP_SYNC_PMATRIX)) return PLL_FAILURE;
    memcpy(p->pmatrix[0], c->pmatrix[0], sizeof(double) * (size_t)e->nmat * e->R * e->S * e->Sp);
    what &= ~(unsigned)PLLHIP_SYNC_PMATRIX;
  }
  if (e->shards.empty()) PLLHIP_TRY(hipSetDevice(e->device));
  if ((what & PLLHIP_SYNC_PMATRIX) && e->pmat_host_dirty && e->nmat)
  {
    if (!flush_pmatrices(p)) return PLL_FAILURE;
    PLLHIP_TRY(hipMemcpyAsync(p->pmatrix[0], e->d_pmat,
                              sizeof(double) * (size_t)e->nmat * e->R * e->S * e->Sp,
                              hipMemcpyDeviceToHost, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    e->pmat_host_dirty = false;
  }
  if (what & PLLHIP_SYNC_CLV)
  {
    const size_t len = (size_t)e->N * e->R * e->Sp;
    for (unsigned i = 0; i < e->nodes; ++i)
    {
      if (i < e->tips && e->coded_tips) continue;   // libpll keeps no CLV for coded tips either
      if (!p->clv[i]) p->clv[i] = static_cast<double *>(pll_aligned_alloc(sizeof(double) * (len ? len : 1), p->alignment));
      if (!p->clv[i]) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate host CLV mirror"); return PLL_FAILURE; }
      if (!pllhip_get_clv(p, i, p->clv[i])) return PLL_FAILURE;
    }
  }
  if (what & PLLHIP_SYNC_SCALERS)
    for (unsigned i = 0; i < e->nscalers; ++i)
    {
      if (!p->scale_buffer[i])
        p->scale_buffer[i] = static_cast<unsigned *>(calloc((e->N ? e->N : 1) * (e->rate_scalers ? (size_t)e->R : 1),
                                                            sizeof(unsigned)));
      if (!p->scale_buffer[i]) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate host scaler mirror"); return PLL_FAILURE; }
      if (!pllhip_get_scaler(p, i, p->scale_buffer[i])) return PLL_FAILURE;
    }
  // a dump of this partition tells the loader to create one with host mirrors again
  if (what & (PLLHIP_SYNC_CLV | PLLHIP_SYNC_SCALERS)) p->attributes |= PLLHIP_ATTRIB_HOST_MIRRORS;
  return PLL_SUCCESS;
}

int pllhip_sync_to_device(pll_partition_t * p, unsigned int what)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty() && (what & PLLHIP_SYNC_PMATRIX) && e->nmat)
  {
    for (pll_partition_t * c : e->shards)
    {
      memcpy(c->pmatrix[0], p->pmatrix[0], sizeof(double) * (size_t)e->nmat * e->R * e->S * e->Sp);
      if (!pllhip_sync_to_device(c, PLLHIP_SYNC_PMATRIX)) return PLL_FAILURE;
    }
    what &= ~(unsigned)PLLHIP_SYNC_PMATRIX;
  }
  if (!e->shards.empty())
    for (pll_partition_t * c : e->shards) push_model(p, c);     // a loader filled the parent's model arrays
  if (e->shards.empty()) PLLHIP_TRY(hipSetDevice(e->device));
  if (e->shards.empty() && !transient_flush_all(e)) return PLL_FAILURE;
  e->eigen_touched = true;                                  // a loader may have filled the model arrays
  e->pmatrix_burst = false;
  if (what & PLLHIP_SYNC_TIPS)
  {
    if (e->coded_tips)
    {
      e->tipmap_codes_uploaded = ~0u;                       // the map itself may be new
      for (unsigned t = 0; t < e->tips; ++t)
        if (p->tipchars && p->tipchars[t] && !upload_tip_codes(p, t)) return PLL_FAILURE;
      invalidate_luts(p);
    }
    p->pattern_weight_sum = 0;
    for (unsigned i = 0; i < p->sites; ++i) p->pattern_weight_sum += p->pattern_weights[i];
    if (!upload_weights(p)) return PLL_FAILURE;
    e->invariant_uploaded = false;
  }
  if ((what & PLLHIP_SYNC_PMATRIX) && e->nmat)
  {
    // the host mirror is the truth now: queued requests are obsolete
    for (unsigned m : e->pend_midx) e->pend_pos[m] = -1;
    e->pend_midx.clear();
    e->pend_t.clear();
    PLLHIP_TRY(hipMemcpyAsync(e->d_pmat, p->pmatrix[0],
                              sizeof(double) * (size_t)e->nmat * e->R * e->S * e->Sp,
                              hipMemcpyHostToDevice, e->stream));
    PLLHIP_TRY(hipStreamSynchronize(e->stream));
    e->pmat_host_dirty = false;
    invalidate_luts(p);
    if (e->d_pfrag)
    {
      hipLaunchKernelGGL(k_s20_pfrag, dim3(e->nmat), dim3(256), 0, e->stream, e->d_pmat, e->d_pfrag, e->R);
      PLLHIP_TRY(hipGetLastError());
    }
  }
  if (what & PLLHIP_SYNC_CLV)
    for (unsigned i = 0; i < e->nodes; ++i)
    {
      if ((i < e->tips && e->coded_tips) || !p->clv[i]) continue;
      if (!pllhip_set_clv(p, i, p->clv[i])) return PLL_FAILURE;
    }
  if (what & PLLHIP_SYNC_SCALERS)
    for (unsigned i = 0; i < e->nscalers; ++i)
      if (p->scale_buffer[i] && !pllhip_set_scaler(p, i, p->scale_buffer[i])) return PLL_FAILURE;
  return PLL_SUCCESS;
}

int pllhip_get_counters(const pll_partition_t * p, pllhip_counters_t * out)
{
  // a sharded partition reports its first shard: every shard does the same calls on its share of the sites
  *out = exec_engine(p)->counters;
  return PLL_SUCCESS;
}

void pllhip_reset_counters(pll_partition_t * p)
{
  engine_of(p)->counters = pllhip_counters_t{};
  for (pll_partition_t * c : engine_of(p)->shards) engine_of(c)->counters = pllhip_counters_t{};
}

int pllhip_profile_partials(pll_partition_t * p, int enable)
{
  Engine * e = engine_of(p);
  for (pll_partition_t * c : e->shards) (void)pllhip_profile_partials(c, enable);
  e->profiling = enable != 0;
  e->prof_used = 0;
  e->prof_bytes = 0.0;
  e->prof_min_bytes = 0.0;
  e->prof_flops = 0.0;
  e->prof_ops = 0;
  return PLL_SUCCESS;
}

int pllhip_profile_read(pll_partition_t * p, pllhip_profile_t * out)
{
  Engine * e = engine_of(p);
  if (!e->shards.empty())
  {
    // bytes and flops add up over the shards; the devices run side by side, so the time is the slowest shard's
    memset(out, 0, sizeof(*out));
    for (pll_partition_t * c : e->shards)
    {
      pllhip_profile_t one;
      if (!pllhip_profile_read(c, &one)) return PLL_FAILURE;
      out->launches = one.launches;
      out->ops = one.ops;
      out->kernel_ms = std::max(out->kernel_ms, one.kernel_ms);
      out->algorithmic_bytes += one.algorithmic_bytes;
      out->algorithmic_flops += one.algorithmic_flops;
      out->minimum_bytes += one.minimum_bytes;
    }
    return PLL_SUCCESS;
  }
  PLLHIP_TRY(hipSetDevice(e->device));
  PLLHIP_TRY(hipStreamSynchronize(e->stream));
  double ms = 0.0;
  for (size_t k = 0; k < e->prof_used; ++k)
  {
    float t = 0.0f;
    PLLHIP_TRY(hipEventElapsedTime(&t, e->prof_events[k].first, e->prof_events[k].second));
    ms += t;
  }
  out->launches = e->prof_used;
  out->ops = e->prof_ops;
  out->kernel_ms = ms;
  out->algorithmic_bytes = e->prof_bytes;
  out->algorithmic_flops = e->prof_flops;
  out->minimum_bytes = e->prof_min_bytes;
  e->prof_used = 0;
  e->prof_bytes = 0.0;
  e->prof_min_bytes = 0.0;
  e->prof_flops = 0.0;
  e->prof_ops = 0;
  return PLL_SUCCESS;
}

const char * pllhip_partials_kernel_name(const pll_partition_t * p)
{
  switch (exec_engine(p)->family)
  {
    case KernelFamily::S4: return "s4-valu";
    case KernelFamily::S20: return "s20-mfma";
    case KernelFamily::S61: return "s61-mfma";
    case KernelFamily::S16: return "s16-mfma";
    default: return "generic";
  }
}

} // extern "C"
