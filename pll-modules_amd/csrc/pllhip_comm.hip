// pllhip_comm.hip -- the reference's parallelism hook on RCCL, and deferred scalar results.
//
// pll-modules synchronises its workers through exactly one callback,
//   void (*parallel_reduce_cb)(void *ctx, double *data, size_t n, int op)
// (src/tree/pll_tree.h:274-276; ops SUM/MAX/MIN = 0/1/2, src/pllmod_common.h:29-31;
// 16 call sites).  Payloads are 8 B .. 8*P B and every
// caller keeps using the result in place: all-reduce semantics.  With one
// process per GPU the transport is an RCCL all-reduce over xGMI.
//
// Two forms:
//   pllhip_reduce_cb     the callback itself, for unmodified pll-modules code: the payload
//                        is on the host (the library calls have returned doubles), so it is
//                        staged through one pinned buffer: H2D, all-reduce, D2H, one wait.
//   pllhip_results_*     deferred results: the reduction kernels of several calls (the
//                        partitions of an evaluation, the trial lengths of a Newton-Raphson
//                        round) leave their totals in device-resident slots, which are
//                        all-reduced IN PLACE on the device; one small kernel publishes
//                        them to mapped host memory and the host waits once.  No host round
//                        trip between the reduction kernel and the collective
//                        (src/tree/treeinfo.c:1058-1067, src/optimize/pll_optimize.c:1270-1286
//                        are the reduces this serves; {df, ddf} travel as one message there too).
#include "engine.h"
#include "pllhip_eval.h"
extern "C" {
#include "host/pllhip_eval_internal.h"
}
#include <rccl/rccl.h>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <limits>
#include <chrono>
#include <cstdlib>
#include <cstdio>

struct pllhip_comm
{
  ncclComm_t comm;
  hipStream_t stream;
  int device, rank, nranks;
  double * d_buf;
  double * h_buf;      // pinned
  size_t cap;
  int aborted;         // comm_abort() ran: every later collective fails at once
};

using namespace pllhip;

static bool nccl_ok(ncclResult_t r, const char * what)
{
  if (r == ncclSuccess) return true;
  set_error(PLL_ERROR_HIP_RUNTIME, "RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
  return false;
}

// ---------------------------------------------------------------------------
// failure handling.  A worker that fails locally must not leave its peers inside a collective:
//   * a failure BEFORE the collective (a deposit that could not be enqueued, an invalid argument) poisons the
//     payload with NaN and still takes part, so that every rank sees NaN and fails in the same call;
//   * a failure IN the collective path (an RCCL / HIP error, or no result within PLLHIP_COLLECTIVE_TIMEOUT_S
//     seconds -- a peer that died leaves the collective's kernel spinning for ever) aborts the communicator
//     (ncclCommAbort: the only call that gets a rank out of a collective its peers never join), sets pll_errno
//     and returns NaN; every later collective on it fails at once with PLL_ERROR_HIP_COMM_ABORTED.
// The process is expected to exit (or to start a fresh child with a new communicator): the library never
// re-executes anything.  PLLHIP_FAULT=deposit@N | collective@N | publish@N (+ PLLHIP_FAULT_RANK=r) injects the
// N-th such event's failure for tests.
// ---------------------------------------------------------------------------
static double collective_timeout()
{
  static const double t = getenv("PLLHIP_COLLECTIVE_TIMEOUT_S") ? atof(getenv("PLLHIP_COLLECTIVE_TIMEOUT_S")) : 120.0;
  return t > 0.0 ? t : 120.0;
}

static void comm_abort(pllhip_comm * c, const char * why)
{
  if (!c || c->aborted) return;
  c->aborted = 1;
  // keep the first error (the cause) if there is one; otherwise say why
  const int code = pll_errno;
  char msg[200];
  snprintf(msg, sizeof(msg), "%s", pll_errmsg);
  if (c->comm) (void)ncclCommAbort(c->comm);
  c->comm = nullptr;
  if (code) set_error(code, "%s [communicator aborted: %s]", msg, why);
  else set_error(PLL_ERROR_HIP_COMM_ABORTED, "communicator aborted: %s", why);
}

static bool comm_usable(const pllhip_comm * c)
{
  if (c && !c->aborted && c->comm) return true;
  set_error(PLL_ERROR_HIP_COMM_ABORTED, "the communicator was aborted after an earlier failure");
  return false;
}

// fault injection: true when this is the N-th event of `kind` on the selected rank
static bool fault_now(const char * kind, int rank)
{
  static const char * spec = getenv("PLLHIP_FAULT");
  static const int only = getenv("PLLHIP_FAULT_RANK") ? atoi(getenv("PLLHIP_FAULT_RANK")) : -1;
  static unsigned long counts[3] = {0, 0, 0};
  if (!spec || (only >= 0 && only != rank)) return false;
  const size_t kl = strlen(kind);
  if (strncmp(spec, kind, kl) != 0 || spec[kl] != '@') return false;
  const unsigned long at = strtoul(spec + kl + 1, nullptr, 10);
  unsigned long & n = counts[kind[0] == 'd' ? 0 : kind[0] == 'c' ? 1 : 2];
  return ++n == at;
}

static ncclRedOp_t nccl_op(int op)
{
  return (op == PLLHIP_REDUCE_MAX) ? ncclMax : (op == PLLHIP_REDUCE_MIN) ? ncclMin : ncclSum;
}

// ---------------------------------------------------------------------------
// deferred results
// ---------------------------------------------------------------------------
constexpr unsigned RESULTS_MAX_PENDING = 64;

struct pllhip_results
{
  pllhip_comm * comm = nullptr;
  int device = 0;
  unsigned nslots = 0;
  double * d_slots = nullptr;                 // device: deposits + in-place all-reduce (communicator mode)
  double * h_slots = nullptr;                 // pinned, device-mapped: what the host reads
  double * hd_slots = nullptr;                // device view of h_slots
  unsigned long long * h_flags = nullptr;     // mapped sequence words: [0] publish kernel, [1 + k] deposit k
  unsigned long long * hd_flags = nullptr;
  unsigned long long seq = 0;
  hipStream_t stream = nullptr;               // collective + publish when deposits come from several streams
  struct Pending { hipStream_t stream; unsigned flag; unsigned long long seq; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> events;
  std::vector<char> deposited;                // per slot, since the last fetch
  unsigned max_pending = 0;                   // sequence words beyond [0]
  bool poisoned = false;                      // a deposit failed since the last fetch: the fetch contributes NaN
};

// slots -> mapped host memory, then the sequence word
__global__ void k_publish_results(const double * src, double * dst, unsigned n,
                                  unsigned long long * flag, unsigned long long seq)
{
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void k_fill_results(double * dst, unsigned n, double v)
{
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = v;
}

// values the host already holds (a partition with ascertainment-bias correction computes its
// scalars in the blocking form: the correction is host arithmetic) -> a slot, and its sequence word
struct HostValues { double v[2 * MAX_TRIAL_LENGTHS]; };

__global__ void k_deposit_values(double * dst, HostValues hv, unsigned n, unsigned long long * flag,
                                 unsigned long long seq)
{
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = hv.v[i];
  __syncthreads();
  if (threadIdx.x == 0 && flag)
  {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int deposit_host_values(pll_partition_t * p, const Engine::Sink & sink, const double * v, unsigned n)
{
  Engine * e = exec_engine(p);
  PLLHIP_TRY(hipSetDevice(e->device));
  HostValues hv;
  memset(&hv, 0, sizeof(hv));
  for (unsigned i = 0; i < n; ++i) hv.v[i] = v[i];
  hipLaunchKernelGGL(k_deposit_values, dim3(1), dim3(64), 0, e->stream, sink.dst, hv, n, sink.flag, sink.seq);
  PLLHIP_TRY(hipGetLastError());
  return PLL_SUCCESS;
}

extern "C" {

int pllhip_comm_get_unique_id(unsigned char id[PLLHIP_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) <= PLLHIP_COMM_ID_BYTES, "id buffer too small");
  ncclUniqueId uid;
  if (!nccl_ok(ncclGetUniqueId(&uid), "ncclGetUniqueId")) return PLL_FAILURE;
  memset(id, 0, PLLHIP_COMM_ID_BYTES);
  memcpy(id, &uid, sizeof(uid));
  return PLL_SUCCESS;
}

pllhip_comm_t * pllhip_comm_create(const unsigned char id[PLLHIP_COMM_ID_BYTES],
                                   int rank, int nranks, int device)
{
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
  pllhip_comm_t * c = new (std::nothrow) pllhip_comm();
  if (!c) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate communicator"); return nullptr; }
  memset(c, 0, sizeof(*c));
  c->device = device; c->rank = rank; c->nranks = nranks; c->cap = 1024;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  bool ok = hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&c->d_buf), c->cap * sizeof(double)), "hipMalloc") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void **>(&c->h_buf), c->cap * sizeof(double),
                                 hipHostMallocDefault), "hipHostMalloc") &&
            nccl_ok(ncclCommInitRank(&c->comm, nranks, uid, rank), "ncclCommInitRank");
  if (!ok) { pllhip_comm_destroy(c); return nullptr; }
  return c;
}

void pllhip_comm_destroy(pllhip_comm_t * c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && !c->aborted) (void)ncclCommDestroy(c->comm);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_buf) (void)hipHostFree(c->h_buf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int pllhip_comm_rank(const pllhip_comm_t * c) { return c ? c->rank : 0; }
int pllhip_comm_size(const pllhip_comm_t * c) { return c ? c->nranks : 1; }

// The reference callback has no error channel and its callers use the payload right
// away: on a HIP / RCCL failure the payload is poisoned with NaN (pll_errno is set), so
// that this rank fails loudly at once (treeinfo asserts lnL < 0, the Newton step rule
// rejects non-finite derivatives) instead of continuing with a local value while its
// peers hold the global one.
void pllhip_reduce_cb(void * ctx, double * data, size_t n, int op)
{
  pllhip_comm_t * c = static_cast<pllhip_comm_t *>(ctx);
  if (!c || !n) return;
  bool ok = comm_usable(c) && hip_ok(hipSetDevice(c->device), "hipSetDevice");
  const ncclRedOp_t rop = nccl_op(op);
  bool in_collective = false;
  for (size_t off = 0; ok && off < n; off += c->cap)
  {
    const size_t m = (n - off < c->cap) ? n - off : c->cap;
    memcpy(c->h_buf, data + off, m * sizeof(double));
    ok = hip_ok(hipMemcpyAsync(c->d_buf, c->h_buf, m * sizeof(double), hipMemcpyHostToDevice, c->stream), "H2D");
    in_collective = ok;
    ok = ok && !(fault_now("collective", c->rank) && (set_error(PLL_ERROR_HIP_RUNTIME, "injected collective failure"), true)) &&
         nccl_ok(ncclAllReduce(c->d_buf, c->d_buf, m, ncclDouble, rop, c->comm, c->stream), "ncclAllReduce") &&
         hip_ok(hipMemcpyAsync(c->h_buf, c->d_buf, m * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H");
    // bounded wait: hipStreamSynchronize would block for ever behind a collective a dead peer never joins
    if (ok)
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;)
      {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { ok = hip_ok(q, "hipStreamQuery"); break; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > collective_timeout())
        {
          set_error(PLL_ERROR_HIP_TIMEOUT, "all-reduce did not complete within %.1f s (peer lost?)", collective_timeout());
          ok = false;
          break;
        }
      }
    }
    if (ok) memcpy(data + off, c->h_buf, m * sizeof(double));
  }
  if (!ok)
  {
    if (in_collective) comm_abort(c, "all-reduce failed");
    for (size_t i = 0; i < n; ++i) data[i] = std::numeric_limits<double>::quiet_NaN();
  }
}

// ---------------------------------------------------------------------------

pllhip_results_t * pllhip_results_create(pllhip_comm_t * comm, unsigned int slots)
{
  if (!slots) { set_error(PLL_ERROR_PARAM_INVALID, "a result group needs at least one slot"); return nullptr; }
  const int device = comm ? comm->device : pllhip_get_device();
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
  pllhip_results_t * rs = new (std::nothrow) pllhip_results();
  if (!rs) { set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate result group"); return nullptr; }
  rs->comm = comm;
  rs->device = device;
  rs->nslots = slots;
  rs->deposited.assign(slots, 0);
  rs->max_pending = std::max(RESULTS_MAX_PENDING, slots);       // every slot may be a deposit of its own
  const size_t flag_bytes = (1 + (size_t)rs->max_pending) * sizeof(unsigned long long);
  bool ok = hip_ok(hipHostMalloc(reinterpret_cast<void **>(&rs->h_slots), slots * sizeof(double),
                                 hipHostMallocMapped), "hipHostMalloc slots") &&
            hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&rs->hd_slots), rs->h_slots, 0), "map slots") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void **>(&rs->h_flags), flag_bytes, hipHostMallocMapped),
                   "hipHostMalloc flags") &&
            hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&rs->hd_flags), rs->h_flags, 0), "map flags") &&
            hip_ok(hipStreamCreateWithFlags(&rs->stream, hipStreamNonBlocking), "hipStreamCreate");
  if (ok)
  {
    memset(rs->h_slots, 0, slots * sizeof(double));
    memset(rs->h_flags, 0, flag_bytes);
  }
  if (ok && comm)
    ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&rs->d_slots), slots * sizeof(double)), "hipMalloc slots") &&
         hip_ok(hipMemset(rs->d_slots, 0, slots * sizeof(double)), "memset slots");
  if (!ok) { pllhip_results_destroy(rs); return nullptr; }
  return rs;
}

void pllhip_results_destroy(pllhip_results_t * rs)
{
  if (!rs) return;
  (void)hipSetDevice(rs->device);
  if (rs->stream) { (void)hipStreamSynchronize(rs->stream); (void)hipStreamDestroy(rs->stream); }
  for (hipEvent_t ev : rs->events) (void)hipEventDestroy(ev);
  if (rs->d_slots) (void)hipFree(rs->d_slots);
  if (rs->h_slots) (void)hipHostFree(rs->h_slots);
  if (rs->h_flags) (void)hipHostFree(rs->h_flags);
  delete rs;
}

} // extern "C"

// where the next reduction of partition `p` leaves `n` totals; remembers what to wait for
static int results_sink(pllhip_results_t * rs, pll_partition_t * p, unsigned slot, unsigned n,
                        Engine::Sink * sink)
{
  if (slot + n > rs->nslots || slot + n < slot)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "result slots %u..%u out of range (%u)", slot, slot + n, rs->nslots);
    return PLL_FAILURE;
  }
  Engine * e = exec_engine(p);
  if (e->device != rs->device)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "partition on device %d, result group on device %d", e->device, rs->device);
    return PLL_FAILURE;
  }
  if (rs->pending.size() >= rs->max_pending)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "more than %u deferred results without a fetch", rs->max_pending);
    return PLL_FAILURE;
  }
  if (fault_now("deposit", rs->comm ? rs->comm->rank : 0))
  {
    set_error(PLL_ERROR_HIP_RUNTIME, "injected deposit failure");
    return PLL_FAILURE;
  }
  pllhip_results::Pending pd;
  pd.stream = e->stream;
  pd.flag = 1 + (unsigned)rs->pending.size();
  pd.seq = ++rs->seq;
  if (rs->comm)
  {
    sink->dst = rs->d_slots + slot;
    sink->flag = nullptr;
  }
  else
  {
    sink->dst = rs->hd_slots + slot;          // straight into mapped memory, with its own sequence word
    sink->flag = rs->hd_flags + pd.flag;
  }
  sink->seq = pd.seq;
  sink->nq = n;
  rs->pending.push_back(pd);
  for (unsigned i = 0; i < n; ++i) rs->deposited[slot + i] = 1;
  return PLL_SUCCESS;
}

extern "C" {

int pllhip_results_edge_loglikelihood(pllhip_results_t * rs, unsigned int slot, pll_partition_t * p,
                                      unsigned int parent_clv_index, int parent_scaler_index,
                                      unsigned int child_clv_index, int child_scaler_index,
                                      unsigned int matrix_index, const unsigned int * freqs_indices)
{
  Engine::Sink sink;
  // a deposit that fails leaves the group poisoned: the fetch then contributes NaN, so that every rank fails together
  if (!results_sink(rs, p, slot, 1, &sink)) { rs->poisoned = true; return PLL_FAILURE; }
  if (p->asc_bias_alloc || is_router(p))
  {
    // the scalar is host arithmetic here (ascertainment-bias correction; sum over the shards of
    // a partition spread over several devices): the blocking form, then the value into the slot
    const double v = pll_compute_edge_loglikelihood(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                                    child_scaler_index, matrix_index, freqs_indices, nullptr);
    if ((!std::isfinite(v) && pll_errno) || !deposit_host_values(p, sink, &v, 1)) { rs->poisoned = true; return PLL_FAILURE; }
    return PLL_SUCCESS;
  }
  const double v = loglikelihood_impl(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                      child_scaler_index, (int)matrix_index, freqs_indices, nullptr, &sink);
  if (v != 0.0) rs->poisoned = true;
  return (v == 0.0) ? PLL_SUCCESS : PLL_FAILURE;
}

void pllhip_results_poison(pllhip_results_t * rs) { if (rs) rs->poisoned = true; }

int pllhip_results_derivatives(pllhip_results_t * rs, unsigned int slot, pll_partition_t * p,
                               int parent_scaler_index, int child_scaler_index,
                               const double * branch_lengths, unsigned int count,
                               const unsigned int * params_indices, const double * sumtable)
{
  Engine::Sink sink;
  if (count > MAX_TRIAL_LENGTHS || !results_sink(rs, p, slot, 2 * count, &sink)) { rs->poisoned = true; return PLL_FAILURE; }
  int rc;
  if (p->asc_bias_alloc || is_router(p))
  {
    double df[MAX_TRIAL_LENGTHS], ddf[MAX_TRIAL_LENGTHS], v[2 * MAX_TRIAL_LENGTHS];
    rc = pllhip_compute_likelihood_derivatives_multi(p, parent_scaler_index, child_scaler_index, branch_lengths,
                                                     count, params_indices, sumtable, df, ddf);
    for (unsigned i = 0; rc && i < count; ++i) { v[2 * i] = df[i]; v[2 * i + 1] = ddf[i]; }
    rc = rc && deposit_host_values(p, sink, v, 2 * count);
  }
  else
    rc = derivatives_impl(p, parent_scaler_index, child_scaler_index, branch_lengths, count, params_indices,
                          sumtable, &sink, nullptr, nullptr);
  if (!rc) rs->poisoned = true;
  return rc;
}

// the collective part of a fetch in communicator mode; false on any failure (pll_errno says which)
static bool fetch_collective(pllhip_results_t * rs, unsigned first, unsigned count, int op, double identity, bool * entered)
{
  // one stream carries the collective: the depositing stream itself when there is only
  // one (a single partition per rank: no event, no cross-stream wait), else the group's
  // stream behind an event of every depositing stream
  hipStream_t run = rs->stream;
  bool single = !rs->pending.empty();
  for (const auto & pd : rs->pending) if (pd.stream != rs->pending[0].stream) single = false;
  if (single) run = rs->pending[0].stream;
  else
  {
    std::vector<hipStream_t> seen;
    for (const auto & pd : rs->pending)
    {
      bool dup = false;
      for (hipStream_t s : seen) if (s == pd.stream) dup = true;
      if (dup) continue;
      if (seen.size() == rs->events.size())
      {
        hipEvent_t ev;
        if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return false;
        rs->events.push_back(ev);
      }
      hipEvent_t ev = rs->events[seen.size()];
      seen.push_back(pd.stream);
      if (!hip_ok(hipEventRecord(ev, pd.stream), "hipEventRecord") ||
          !hip_ok(hipStreamWaitEvent(run, ev, 0), "hipStreamWaitEvent")) return false;
    }
  }
  if (rs->poisoned)
  {
    // a deposit failed on this rank: NaN into every slot, so that the peers fail in this very call as well
    hipLaunchKernelGGL(k_fill_results, dim3(1), dim3(64), 0, run, rs->d_slots + first, count,
                       std::numeric_limits<double>::quiet_NaN());
    if (!hip_ok(hipGetLastError(), "poison")) return false;
  }
  else
    // slots nobody deposited to (partitions another worker owns) take the identity
    for (unsigned i = 0; i < count; )
    {
      if (rs->deposited[first + i]) { ++i; continue; }
      unsigned j = i;
      while (j < count && !rs->deposited[first + j]) ++j;
      hipLaunchKernelGGL(k_fill_results, dim3(1), dim3(64), 0, run, rs->d_slots + first + i, j - i, identity);
      if (!hip_ok(hipGetLastError(), "fill")) return false;
      i = j;
    }
  const unsigned long long seq = ++rs->seq;
  *entered = true;
  if (fault_now("collective", rs->comm->rank))
  {
    set_error(PLL_ERROR_HIP_RUNTIME, "injected collective failure");
    return false;
  }
  if (!nccl_ok(ncclAllReduce(rs->d_slots + first, rs->d_slots + first, count, ncclDouble, nccl_op(op),
                             rs->comm->comm, run), "ncclAllReduce"))
    return false;
  // (publish@N: the result never reaches the host, as if the collective in front of it never completed)
  const bool lose = fault_now("publish", rs->comm->rank);
  hipLaunchKernelGGL(k_publish_results, dim3(1), dim3(64), 0, run, rs->d_slots + first, rs->hd_slots + first,
                     count, lose ? rs->hd_flags + rs->max_pending : rs->hd_flags, lose ? 0ULL : seq);
  if (!hip_ok(hipGetLastError(), "publish")) return false;
  if (lose)
  {
    // the stream drains, the flag never comes: what a lost peer looks like from here
    const double t = collective_timeout();
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < t) { }
    set_error(PLL_ERROR_HIP_TIMEOUT, "no result after %.1f s: a collective did not complete (peer lost?)", t);
    return false;
  }
  return wait_sequence(run, rs->h_flags, seq, collective_timeout()) != 0;
}

int pllhip_results_fetch(pllhip_results_t * rs, unsigned int first, unsigned int count, int op, double * out)
{
  const double nan = std::numeric_limits<double>::quiet_NaN();
  int rc = PLL_SUCCESS;
  if (first + count > rs->nslots || !count)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "result slots %u..%u out of range (%u)", first, first + count, rs->nslots);
    rc = PLL_FAILURE;
  }
  else if (!hip_ok(hipSetDevice(rs->device), "hipSetDevice")) rc = PLL_FAILURE;
  const double identity = (op == PLLHIP_REDUCE_MAX) ? -INFINITY : (op == PLLHIP_REDUCE_MIN) ? INFINITY : 0.0;
  if (rc && !rs->comm)
  {
    // every deposit went straight to mapped memory with a sequence word of its own
    for (const auto & pd : rs->pending)
      if (!wait_sequence(pd.stream, rs->h_flags + pd.flag, pd.seq)) rc = PLL_FAILURE;
    for (unsigned i = 0; i < count; ++i)
      out[i] = rs->deposited[first + i] ? rs->h_slots[first + i] : identity;
    if (rs->poisoned)
    {
      if (!pll_errno) set_error(PLL_ERROR_HIP_RUNTIME, "a deferred result could not be enqueued");
      rc = PLL_FAILURE;
    }
  }
  else if (rc)
  {
    bool entered = false;
    if (!comm_usable(rs->comm)) rc = PLL_FAILURE;
    else if (!fetch_collective(rs, first, count, op, identity, &entered))
    {
      // before the collective was enqueued the peers are still waiting for this rank, inside it they wait for a
      // rank that will not come back: either way only an abort gets everybody out
      comm_abort(rs->comm, entered ? "the all-reduce of deferred results failed" : "a rank could not enter the all-reduce");
      rc = PLL_FAILURE;
    }
    else
    {
      for (unsigned i = 0; i < count; ++i) out[i] = rs->h_slots[first + i];
      if (rs->poisoned)
      {
        if (!pll_errno) set_error(PLL_ERROR_HIP_RUNTIME, "a deferred result could not be enqueued");
        rc = PLL_FAILURE;
      }
    }
  }
  // whatever happened, the group is ready for the next round of deposits
  rs->pending.clear();
  rs->poisoned = false;
  std::fill(rs->deposited.begin(), rs->deposited.end(), 0);
  if (!rc && out && first + count <= rs->nslots)
    for (unsigned i = 0; i < count; ++i) out[i] = nan;
  return rc;
}

} // extern "C"

// ---------------------------------------------------------------------------
// the evaluation driver (include/pllhip_eval.h) on deferred results
// ---------------------------------------------------------------------------
static int fused_edge(void * rs, unsigned int slot, pll_partition_t * p, unsigned int pc, int psc,
                      unsigned int cc, int csc, unsigned int m, const unsigned int * f)
{
  return pllhip_results_edge_loglikelihood(static_cast<pllhip_results_t *>(rs), slot, p, pc, psc, cc, csc, m, f);
}

static int fused_deriv(void * rs, unsigned int slot, pll_partition_t * p, int psc, int csc,
                       const double * t, unsigned int count, const unsigned int * params, const double * sumtable)
{
  return pllhip_results_derivatives(static_cast<pllhip_results_t *>(rs), slot, p, psc, csc, t, count, params,
                                    sumtable);
}

static int fused_fetch(void * rs, unsigned int first, unsigned int count, int op, double * out)
{
  return pllhip_results_fetch(static_cast<pllhip_results_t *>(rs), first, count, op, out);
}

static void fused_destroy(void * rs) { pllhip_results_destroy(static_cast<pllhip_results_t *>(rs)); }
static void fused_poison(void * rs) { pllhip_results_poison(static_cast<pllhip_results_t *>(rs)); }

extern "C" int pllhip_eval_attach_comm(struct pllhip_eval * ev, pllhip_comm_t * comm)
{
  if (!ev) { set_error(PLL_ERROR_PARAM_INVALID, "no evaluator"); return PLL_FAILURE; }
  pllhip_results_t * rs = pllhip_results_create(comm, ev->nparts * 2 * PLLHIP_EVAL_MAX_TRIALS);
  if (!rs) return PLL_FAILURE;
  pllhip_eval_fused_t table;
  table.results = rs;
  table.edge_loglikelihood = fused_edge;
  table.derivatives = fused_deriv;
  table.fetch = fused_fetch;
  table.destroy = fused_destroy;
  table.poison = fused_poison;
  pllhip_eval_set_fused(ev, &table);
  // control decisions of the driver (MIN / MAX of host values) go through the callback
  if (comm) pllhip_eval_set_parallel_context(ev, comm, pllhip_reduce_cb);
  return PLL_SUCCESS;
}
