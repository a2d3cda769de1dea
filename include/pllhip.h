/*
 * pllhip.h -- engine-specific additions to the libpll-2 style C ABI of
 * include/pll.h.  Nothing here exists in the reference; these entry points
 * cover what a device-resident engine needs on top of the reference interface:
 *
 *   - device selection and introspection,
 *   - explicit materialisation of device-resident arrays into the host mirrors
 *     that pll-modules' msa/binary code reads (partition->clv[i],
 *     partition->scale_buffer[i], partition->pmatrix[i]):
 *     src/binary/binary_io_operations.c:286-296, src/msa/pll_msa.c:114-124,
 *     test/src/optimize/blopt-minimal.c:96 (pll_show_pmatrix),
 *   - a ready-made implementation of the reference's only parallelism hook,
 *       void (*parallel_reduce_cb)(void *ctx, double *data, size_t n, int op)
 *     (src/tree/pll_tree.h:274-276, ops SUM/MAX/MIN = 0/1/2 at
 *     src/pllmod_common.h:29-31), backed by RCCL over xGMI with one process
 *     per GPU.
 */
#ifndef PLLHIP_H_INCLUDED
#define PLLHIP_H_INCLUDED

#include "pll.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what pllhip_sync_to_host materialises */
#define PLLHIP_SYNC_PMATRIX  (1 << 0)   /* all P-matrices -> partition->pmatrix   */
#define PLLHIP_SYNC_CLV      (1 << 1)   /* all CLVs       -> partition->clv[i]    */
#define PLLHIP_SYNC_SCALERS  (1 << 2)   /* all scalers    -> partition->scale_buffer[i] */
#define PLLHIP_SYNC_TIPS     (1 << 3)   /* pllhip_sync_to_device only: tip codes, tip map, pattern weights, invariant sites */
#define PLLHIP_SYNC_ALL      15

/* pll_partition_create attribute (above PLL_ATTRIB_MASK): allocate the host mirrors
   partition->clv[i] / scale_buffer[i] up front, so that code which FILLS them -- the
   reference's binary loader, src/binary/pll_binary.c:346-500 -- finds memory there;
   pllhip_sync_to_device() then moves the contents to the GPU.  pllhip_sync_to_host()
   sets the bit on the partition, so a dump carries it into the file and the loader
   passes it back to pll_partition_create. */
#define PLLHIP_ATTRIB_HOST_MIRRORS (1u << 30)

/* reduce-callback operation codes (src/pllmod_common.h:29-31) */
#define PLLHIP_REDUCE_SUM 0
#define PLLHIP_REDUCE_MAX 1
#define PLLHIP_REDUCE_MIN 2

/* number of visible HIP devices (0 if none); never initialises a context */
PLL_EXPORT int pllhip_device_count(void);

/* device used by partitions created afterwards on this thread (default 0,
   or the value of the PLLHIP_DEVICE environment variable) */
PLL_EXPORT int pllhip_set_device(int device);
PLL_EXPORT int pllhip_get_device(void);

/* gfx architecture name of a device, e.g. "gfx950" */
PLL_EXPORT int pllhip_device_arch(int device, char * out, size_t out_len);

/* host-only: eigen-decomposition of the reversible rate matrix built from
   `subst_params` (upper triangle, row-major) and `frequencies`, normalised to
   mean rate 1.  eigenvecs[i*Sp+k] = V[i][k], inv_eigenvecs[k*Sp+j] = V^-1[k][j].
   This is what pll_update_prob_matrices runs when eigen_decomp_valid[i] == 0. */
PLL_EXPORT int pllhip_eigen_decompose(unsigned int states, unsigned int states_padded,
                                      const double * subst_params,
                                      const double * frequencies,
                                      double * eigenvecs, double * inv_eigenvecs,
                                      double * eigenvals);

/* copy device-resident arrays into the host mirrors of the partition
   (allocating partition->clv[i] / scale_buffer[i] on first use) */
PLL_EXPORT int pllhip_sync_to_host(pll_partition_t * partition, unsigned int what);

/* the opposite direction: host mirrors (filled by a checkpoint loader) -> device.
   CLVs / scalers whose mirror is NULL are skipped. */
PLL_EXPORT int pllhip_sync_to_device(pll_partition_t * partition, unsigned int what);

/* single-array variants, caller-provided output.  Layouts are the partition's:
   CLV [site][rate][states_padded]; a tip stored as codes is expanded to 0/1. */
PLL_EXPORT int pllhip_get_clv(pll_partition_t * partition, unsigned int clv_index,
                              double * out);
PLL_EXPORT int pllhip_get_scaler(pll_partition_t * partition, unsigned int scaler_index,
                                 unsigned int * out);
PLL_EXPORT int pllhip_get_sumtable(pll_partition_t * partition,
                                   const double * sumtable_key, double * out);

/* host -> device for an inner CLV / scaler (checkpoint restore path of
   src/binary; also lets tests inject states) */
PLL_EXPORT int pllhip_set_clv(pll_partition_t * partition, unsigned int clv_index,
                              const double * clv);
PLL_EXPORT int pllhip_set_scaler(pll_partition_t * partition, unsigned int scaler_index,
                                 const unsigned int * scaler);

/* block until all work queued on the partition's stream has finished */
PLL_EXPORT int pllhip_synchronize(pll_partition_t * partition);

/* the partition's HIP stream (hipStream_t as void*), for callers that time
   kernels with events or enqueue their own work */
PLL_EXPORT void * pllhip_stream(pll_partition_t * partition);

/* per-partition work counters (engine-side analogue of treeinfo->counter,
   src/tree/treeinfo.c:1017) */
typedef struct pllhip_counters
{
  unsigned long long partial_ops;        /* operations executed                */
  unsigned long long partial_launches;   /* kernel launches for them           */
  unsigned long long site_updates;       /* ops * sites * rate_cats            */
  unsigned long long pmatrix_updates;      /* matrices requested               */
  unsigned long long pmatrix_launches;     /* kernel launches that served them */
  unsigned long long lnl_calls;
  unsigned long long sumtable_calls;
  unsigned long long derivative_calls;
  unsigned long long model_uploads;      /* host->device re-syncs of model state */
} pllhip_counters_t;

PLL_EXPORT int pllhip_get_counters(const pll_partition_t * partition,
                                   pllhip_counters_t * out);
PLL_EXPORT void pllhip_reset_counters(pll_partition_t * partition);

/* live timing of the pll_update_partials kernel launches with HIP events on the
   partition's stream.  While enabled, every launch is bracketed by two events;
   pllhip_profile_read() synchronises, sums the elapsed times, reports them with
   the algorithmic byte count of those launches and
   resets the accumulators. */
typedef struct pllhip_profile
{
  unsigned long long launches;
  unsigned long long ops;
  double kernel_ms;
  double algorithmic_bytes;
  double algorithmic_flops;   /* 2*S*S per non-tip child matvec + S products, per site-update */
} pllhip_profile_t;

PLL_EXPORT int pllhip_profile_partials(pll_partition_t * partition, int enable);
PLL_EXPORT int pllhip_profile_read(pll_partition_t * partition, pllhip_profile_t * out);

/* kernel family actually used for pll_update_partials on this partition:
   "s4-valu", "s20-mfma", "generic" ... (for tests that must prove the
   specialised path ran) */
PLL_EXPORT const char * pllhip_partials_kernel_name(const pll_partition_t * partition);

/* ---- multi-GPU: one process per GPU, RCCL all-reduce ---------------- */

#define PLLHIP_COMM_ID_BYTES 128

/* rank 0 creates an id and ships it to the other ranks by any side channel
   (bench.py uses the torch.distributed store) */
PLL_EXPORT int pllhip_comm_get_unique_id(unsigned char id[PLLHIP_COMM_ID_BYTES]);

typedef struct pllhip_comm pllhip_comm_t;

PLL_EXPORT pllhip_comm_t * pllhip_comm_create(const unsigned char id[PLLHIP_COMM_ID_BYTES],
                                              int rank, int nranks, int device);
PLL_EXPORT void pllhip_comm_destroy(pllhip_comm_t * comm);

/* drop-in value for treeinfo's parallel_reduce_cb with ctx = pllhip_comm_t* */
PLL_EXPORT void pllhip_reduce_cb(void * ctx, double * data, size_t n, int op);

#ifdef __cplusplus
}
#endif

#endif /* PLLHIP_H_INCLUDED */
