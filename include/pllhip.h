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

/* ---- one partition over several GPUs (engine-internal sharding, SURVEY.md 8e topology i) ----
 * Partitions created afterwards on this thread are split into `count` contiguous site ranges,
 * one per device (devices[i]; NULL: device i modulo the visible ones): every device holds all
 * nodes' CLVs of its range and a replica of the model; pll_update_partials /
 * pll_update_prob_matrices / pll_update_sumtable fan out without communication, the scalar
 * calls add the devices' sums on the host in a fixed order.  The caller -- an unmodified
 * pll-modules client with ONE treeinfo and no parallel_reduce_cb -- sees one pll_partition_t.
 * count <= 1 switches it off.  The environment variable PLLHIP_SHARD_DEVICES="0,1,2,3" does the
 * same for clients that cannot call this.  Partitions with fewer than 64 sites per device, and
 * partitions with ascertainment-bias correction, stay on one device. */
PLL_EXPORT int pllhip_set_sharding(unsigned int count, const int * devices);
/* number of devices `partition` is spread over (1: an ordinary partition) */
PLL_EXPORT unsigned int pllhip_shard_count(const pll_partition_t * partition);

/* gfx architecture name of a device, e.g. "gfx950" */
PLL_EXPORT int pllhip_device_arch(int device, char * out, size_t out_len);

/* host-only: eigen-decomposition of the reversible rate matrix built from
   `subst_params` (upper triangle, row-major) and `frequencies`, normalised to
   mean rate 1, Q = V L V^-1.  Storage follows libpll-2: inv_eigenvecs[i*Sp+k] = V[i][k],
   eigenvecs[k*Sp+j] = V^-1[k][j], i.e. P(t) = inv_eigenvecs * diag(exp(L t)) * eigenvecs.
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
  unsigned long long derivative_calls;   /* sumtable scans                     */
  unsigned long long derivative_points;  /* trial branch lengths evaluated by them */
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
  double minimum_bytes;       /* what the schedule has to move: as algorithmic_bytes, but a child vector (and its
                                 scaler counts) that an operation chain hands over in registers is not read */
} pllhip_profile_t;

PLL_EXPORT int pllhip_profile_partials(pll_partition_t * partition, int enable);

/* PLL_ATTRIB_SITE_REPEATS (libpll-2's site repeats; the reference's test harness selects it,
   test/src/common.c:31): what the engine did with it since the partition was created.  The vector of a cherry
   (a tip x tip operation) is computed per class of sites -- a pair of tip codes -- and not per site, and so is the
   vector of every node whose two children are known per class (4- and 20-state families, coded tips). */
typedef struct pllhip_repeat_stats
{
  unsigned long long cherries;         /* operations kept per class (the name is the first step's: cherries only) */
  unsigned long long classes;          /* classes of those operations, summed */
  unsigned long long sites;            /* sites they cover, summed (classes / sites = the share computed) */
  unsigned long long expansions;       /* class nodes expanded to the site-indexed form on demand */
} pllhip_repeat_stats_t;
PLL_EXPORT int pllhip_repeat_stats(const pll_partition_t * partition, pllhip_repeat_stats_t * out);
PLL_EXPORT int pllhip_profile_read(pll_partition_t * partition, pllhip_profile_t * out);

/* Evaluate-only traversals.  The model-parameter optimisers of pll-modules evaluate the whole tree after every
   parameter poke (src/algorithm/algo_callback.c:338, 465, 568, 678: pllmod_treeinfo_compute_loglh(treeinfo, 0);
   src/optimize/opt_algorithms.c:734-773: nmax + 1 of them per L-BFGS-B iteration): every vector is recomputed by
   the next evaluation and never read in between, yet two thirds of what such an evaluation moves are stores.
   While the mode is on, an operation list with the shape of a tree traversal (pll_update_partials /
   pllhip_update_partials_batch, resident schedules of the 4-, 2..32- and 20-state families) hands the vectors
   inside its operation chains on in registers WITHOUT storing them; the last vector of every chain and all scaler
   counts are stored as always.  Nothing observable changes: a vector that was not stored stays recomputable (the
   engine keeps its operation) and is stored
     - for the first reader that needs it (a later operation list, an edge / root log-likelihood, a sumtable,
       pllhip_get_clv, pllhip_sync_to_host, ...), and
     - before one of its inputs changes (a P-matrix it was computed with, a tip, a child vector or scaler buffer
       that a later list overwrites),
   with the very operations that made it, so every later result is bit-identical to the mode being off.
   pllhip_discard_transient declares the vectors that were not stored dead (a caller that is about to change the
   model and evaluate the whole tree again: nothing is recomputed for the P-matrix updates that follow);
   reading one of them afterwards without recomputing it is the caller's error, as after any invalidation.
   PLLHIP_TRANSIENT=1 in the environment switches the mode on for every partition (the test suite under it). */
typedef struct pllhip_transient_stats
{
  unsigned long long skipped;          /* vectors a traversal did not store */
  unsigned long long materialized;     /* ... that were recomputed and stored for a reader or before a change */
  unsigned long long discarded;        /* ... that were declared dead or overwritten before anybody asked */
} pllhip_transient_stats_t;
PLL_EXPORT int pllhip_set_transient(pll_partition_t * partition, int enable);
PLL_EXPORT int pllhip_discard_transient(pll_partition_t * partition);
PLL_EXPORT int pllhip_transient_stats(const pll_partition_t * partition, pllhip_transient_stats_t * out);

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

PLL_EXPORT int pllhip_comm_rank(const pllhip_comm_t * comm);
PLL_EXPORT int pllhip_comm_size(const pllhip_comm_t * comm);

/* drop-in value for treeinfo's parallel_reduce_cb with ctx = pllhip_comm_t*.  The payload is
   a host array (the library calls before it have returned doubles), staged through one pinned
   buffer.  On a HIP / RCCL failure the payload is set to NaN and pll_errno is set: the
   callback has no error channel, and a rank must not continue with its local value. */
PLL_EXPORT void pllhip_reduce_cb(void * ctx, double * data, size_t n, int op);

/* ---- several trial branch lengths per sumtable scan ------------------------
 * pll_compute_likelihood_derivatives at `count` (1..8) branch lengths in ONE pass over the
 * sumtable (src/optimize/pll_optimize.c:1223-1287 scans it once per Newton-Raphson
 * iteration).  d_f[i], dd_f[i] are bit-identical to what the single-length call returns
 * for branch_lengths[i], whatever else shares the launch. */
PLL_EXPORT int pllhip_compute_likelihood_derivatives_multi(pll_partition_t * partition,
                                                           int parent_scaler_index,
                                                           int child_scaler_index,
                                                           const double * branch_lengths,
                                                           unsigned int count,
                                                           const unsigned int * params_indices,
                                                           const double * sumtable,
                                                           double * d_f, double * dd_f);

/* how many trial lengths one scan of this partition's sumtable evaluates at (about) the
   price of one: 4 where the scan runs on the matrix cores (20- and 61-state families: the
   lengths are rows of an MFMA operand and the scan stays HBM-bound), 1 elsewhere (the
   4-state scan is bound by its per-site divisions).  Callers that speculate on trial
   lengths (include/pllhip_eval.h) use it to decide whether speculation is free. */
PLL_EXPORT unsigned int pllhip_free_trial_lengths(const pll_partition_t * partition);

/* Newton-Raphson on one branch, entirely on the device: the loop the reference runs around
   pll_compute_likelihood_derivatives (src/optimize/opt_algorithms.c:133-261: evaluate {f, f'} at the iterate,
   bracket, clamp the step to +-bl_max / max_newton and to the bracket, stop at |f| or |step| < tolerance; the
   target function is src/optimize/pll_optimize.c:1223-1287) for ONE partition whose sumtable is current
   (pll_update_sumtable), in ONE launch: scan, in-launch reduction, step rule, next scan.  The iterates are the
   host loop's, bit for bit (same scan, same summation order, the same fp64 expressions without contraction).
   Returns PLL_SUCCESS with *length = the length the loop ended at and *iterations = scans made; `trail`
   (NULL, or room for 96 doubles) receives the iterate after every scan.  PLL_FAILURE with pll_errno =
   PLLHIP_ERROR_NEWTON_LIMIT (more than max_newton iterations) / PLLHIP_ERROR_NEWTON_DERIVATIVES (a non-finite
   derivative) -- the reference's two failure modes -- or PLLHIP_ERROR_NEWTON_UNSUPPORTED: this partition cannot
   run the loop on the device (4-state / generic kernel family, ascertainment-bias correction, a partition spread
   over devices, a scan grid larger than the chip holds at once); the caller then iterates itself.
   PLLHIP_ERROR_NEWTON_STUCK: the workgroups of the loop wait for one another inside the launch, so all of them
   have to be on the chip at once; the grid is sized for a device this partition has to itself, and other work on
   the device (another process, another stream) can keep a workgroup out.  Every wait is bounded: the launch then
   ends with this code after the bound (about a second), the engine's reduction state is reset, nothing was
   changed -- the caller iterates itself (pllhip_eval does, and stops asking for the device loop). */
#define PLLHIP_ERROR_NEWTON_LIMIT        910
#define PLLHIP_ERROR_NEWTON_DERIVATIVES  911
#define PLLHIP_ERROR_NEWTON_UNSUPPORTED  912
#define PLLHIP_ERROR_NEWTON_STUCK        913
PLL_EXPORT int pllhip_newton_branch(pll_partition_t * partition,
                                    int parent_scaler_index, int child_scaler_index,
                                    const unsigned int * params_indices, const double * sumtable,
                                    double start, double bl_min, double bl_max, double tolerance,
                                    unsigned int max_newton,
                                    double * length, unsigned int * iterations, double * trail);

/* The same loop for SEVERAL partitions that share the branch length -- linked lengths, or scaled ones (partition p
   sees length_scalers[p] * x; NULL: all 1) --: the reference's multi-partition derivative function
   (src/optimize/pll_optimize.c:1223-1287: f = sum s_p f_p, f' = sum s_p^2 f'_p, added in partition order) inside
   the loop.  Every partition runs its own instance of the loop on its own stream (its family's kernel, its own scan
   grid: its totals are those of its blocking derivative call); the instances meet on the device after every scan.
   All partitions live on ONE device and none is remote (a sum over workers needs the host loop); at most 8.
   The partitions run as ONE launch (every partition a run of its workgroups, every family its own loop) when their scan
   grids fit the chip together under that kernel's occupancy; else one launch per partition.
   Same results and error codes as pllhip_newton_branch; PLLHIP_ERROR_NEWTON_UNSUPPORTED also when the partitions'
   scan grids do not fit the chip together, or -- one launch per partition -- the process was not started with
   GPU_MAX_HW_QUEUES >= 8 in its environment: those launches wait for one another on the device, the HIP runtime runs streams that share a
   hardware queue one after the other, and it has four queues unless told otherwise before its first call (the library
   does not ask for more: they slow evaluations of many partitions down).  The caller iterates from the host then. */
PLL_EXPORT int pllhip_newton_branch_multi(pll_partition_t * const * partitions, unsigned int count,
                                          int parent_scaler_index, int child_scaler_index,
                                          const unsigned int * const * params_indices,
                                          const double * const * sumtables, const double * length_scalers,
                                          double start, double bl_min, double bl_max, double tolerance,
                                          unsigned int max_newton,
                                          double * length, unsigned int * iterations, double * trail);

/* pll_update_partials for several partitions that are evaluated on ONE tree: the result is what
   pll_update_partials(partitions[i], operations, count) stores for every non-NULL partitions[i], bit for
   bit.  pll-modules walks the partitions of an analysis one after the other
   (src/tree/treeinfo.c:1020-1056, src/optimize/pll_optimize.c:748-775); partitions of one kernel family on
   one device share their launches here (the chains of a round of every partition are grid rows of one
   launch), which is what keeps data sets with one small partition per gene -- and the per-GPU share of a
   partitioned analysis on eight GPUs -- off the launch-latency floor.  NULL entries (partitions another
   worker owns, src/tree/treeinfo.c:1024-1031) are skipped.  PLLHIP_BATCH=0: per-partition calls. */
PLL_EXPORT int pllhip_update_partials_batch(pll_partition_t * const * partitions,
                                            unsigned int partition_count,
                                            const pll_operation_t * operations,
                                            unsigned int count);

/* ---- deferred scalar results -----------------------------------------------
 * pll_compute_edge_loglikelihood and pll_compute_likelihood_derivatives hand a double back
 * to the host: one wait per call and partition, and with several workers a host-side
 * reduce after it.  A result group collects the totals of several such computations -- the
 * partitions of an evaluation (src/tree/treeinfo.c:1040-1067), the trial lengths of a
 * Newton-Raphson round (src/optimize/pll_optimize.c:1240-1286) -- in device-resident slots:
 * the pllhip_results_* calls only enqueue work; pllhip_results_fetch() all-reduces the slots
 * in place over the communicator (when there is one), publishes them to mapped host memory
 * and waits ONCE.  Slots nobody deposited to since the last fetch (partitions another worker
 * owns) count as the identity of `op`.  One fetch completes all deposits made since the
 * previous one.  Same numbers, bit for bit, as the blocking calls. */
typedef struct pllhip_results pllhip_results_t;

PLL_EXPORT pllhip_results_t * pllhip_results_create(pllhip_comm_t * comm /* or NULL */,
                                                    unsigned int slots);
PLL_EXPORT void pllhip_results_destroy(pllhip_results_t * results);

/* 1 slot: the edge log-likelihood of `partition` */
PLL_EXPORT int pllhip_results_edge_loglikelihood(pllhip_results_t * results, unsigned int slot,
                                                 pll_partition_t * partition,
                                                 unsigned int parent_clv_index, int parent_scaler_index,
                                                 unsigned int child_clv_index, int child_scaler_index,
                                                 unsigned int matrix_index,
                                                 const unsigned int * freqs_indices);

/* 2 * count slots: d_f[0], dd_f[0], d_f[1], dd_f[1], ... */
PLL_EXPORT int pllhip_results_derivatives(pllhip_results_t * results, unsigned int slot,
                                          pll_partition_t * partition,
                                          int parent_scaler_index, int child_scaler_index,
                                          const double * branch_lengths, unsigned int count,
                                          const unsigned int * params_indices,
                                          const double * sumtable);

/* out[i] = reduce over all workers of slot first + i; NaN in every out[i] and PLL_FAILURE on error.
   Failure handling (no worker is left inside a collective; the process is expected to exit or to start a
   fresh child -- the library never re-executes anything):
     * a deposit that failed on this worker (or pllhip_results_poison) makes the fetch contribute NaN to
       every slot and still take part in the all-reduce: every worker gets NaN and PLL_FAILURE in this call;
     * an RCCL / HIP error in the collective path, or no result within PLLHIP_COLLECTIVE_TIMEOUT_S seconds
       (default 120: a peer that died never joins), aborts the communicator (ncclCommAbort), sets pll_errno
       (PLL_ERROR_HIP_RUNTIME / PLL_ERROR_HIP_TIMEOUT) and returns NaN; every later collective on that
       communicator fails at once with PLL_ERROR_HIP_COMM_ABORTED.  pllhip_reduce_cb behaves the same way.
   The group itself is reset on every path and can take new deposits. */
PLL_EXPORT int pllhip_results_fetch(pllhip_results_t * results, unsigned int first,
                                    unsigned int count, int op, double * out);

/* this worker cannot contribute to the pending fetch (a local failure outside the deposits) */
PLL_EXPORT void pllhip_results_poison(pllhip_results_t * results);

/* give a pllhip_eval driver (include/pllhip_eval.h) a result group over `comm` (NULL: this
   process only) sized for its partitions; the driver then reduces its lnL and
   {df, ddf} on the device */
struct pllhip_eval;
PLL_EXPORT int pllhip_eval_attach_comm(struct pllhip_eval * ev, pllhip_comm_t * comm);

#ifdef __cplusplus
}
#endif

#endif /* PLLHIP_H_INCLUDED */
