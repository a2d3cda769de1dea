// pllhip_comm.hip -- the reference's parallelism hook, implemented on RCCL.
//
// pll-modules synchronises its workers through exactly one callback,
//   void (*parallel_reduce_cb)(void *ctx, double *data, size_t n, int op)
// (src/tree/pll_tree.h:274-276; ops SUM/MAX/MIN = 0/1/2, src/pllmod_common.h:29-31;
// 16 call sites).  Payloads are 8 B .. 8*P B and every
// caller keeps using the result in place: all-reduce semantics.  With one
// process per GPU the natural transport is an RCCL all-reduce over xGMI; the
// call is latency-bound, so the staging buffers are allocated once and the
// host blocks on one stream synchronise per call.
#include "engine.h"
#include <rccl/rccl.h>
#include <cstring>

struct pllhip_comm
{
  ncclComm_t comm;
  hipStream_t stream;
  int device, rank, nranks;
  double * d_buf;
  double * h_buf;      // pinned
  size_t cap;
};

using namespace pllhip;

static bool nccl_ok(ncclResult_t r, const char * what)
{
  if (r == ncclSuccess) return true;
  set_error(PLL_ERROR_HIP_RUNTIME, "RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
  return false;
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
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_buf) (void)hipHostFree(c->h_buf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void pllhip_reduce_cb(void * ctx, double * data, size_t n, int op)
{
  pllhip_comm_t * c = static_cast<pllhip_comm_t *>(ctx);
  if (!c || !n) return;
  (void)hipSetDevice(c->device);
  const ncclRedOp_t rop = (op == PLLHIP_REDUCE_MAX) ? ncclMax : (op == PLLHIP_REDUCE_MIN) ? ncclMin : ncclSum;
  for (size_t off = 0; off < n; off += c->cap)
  {
    const size_t m = (n - off < c->cap) ? n - off : c->cap;
    memcpy(c->h_buf, data + off, m * sizeof(double));
    if (!hip_ok(hipMemcpyAsync(c->d_buf, c->h_buf, m * sizeof(double), hipMemcpyHostToDevice, c->stream), "H2D") ||
        !nccl_ok(ncclAllReduce(c->d_buf, c->d_buf, m, ncclDouble, rop, c->comm, c->stream), "ncclAllReduce") ||
        !hip_ok(hipMemcpyAsync(c->h_buf, c->d_buf, m * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H") ||
        !hip_ok(hipStreamSynchronize(c->stream), "sync"))
      return;   // pll_errno is set; the reference callback has no error channel
    memcpy(data + off, c->h_buf, m * sizeof(double));
  }
}

} // extern "C"
