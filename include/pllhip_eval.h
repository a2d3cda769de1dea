/*
 * pllhip_eval.h -- a small C driver on top of include/pll.h for machines where
 * pll-modules itself is not available (the GPU box): tree + partitions +
 * validity flags -> likelihood evaluations and Newton-Raphson branch-length
 * optimisation, issuing the same libpll call sequences as the reference's
 * upper layers.  It is NOT a re-implementation of pll-modules: clients that
 * have pll-modules link it unchanged against libpll_hip.so (INTEGRATION.md,
 * tests/test_dropin_modules.py); this driver exists so that those call
 * patterns can be exercised, tested and timed where the reference cannot go.
 *
 * What each entry point mirrors (behaviour, not code):
 *   pllhip_eval_loglh              treeinfo_compute_loglh, src/tree/treeinfo.c:946-1079
 *                                  (full / partial post-order traversal, per-partition
 *                                  loop that skips NULL = remote partitions, SUM reduce)
 *   pllhip_eval_invalidate_*       src/tree/treeinfo.c:872-944 (one CLV slot per inner
 *                                  node: validating one direction invalidates the other two)
 *   pllhip_eval_optimize_branches  pllmod_opt_optimize_branch_lengths_local_multi with
 *                                  linked branch lengths and PLLMOD_OPT_BLO_NEWTON_FAST,
 *                                  src/optimize/pll_optimize.c:1395-1951; Newton-Raphson
 *                                  step rule of pllmod_opt_minimize_newton_multi,
 *                                  src/optimize/opt_algorithms.c:133-261
 *   pllhip_eval_spr_round          pllmod_algo_spr_round, src/algorithm/algo_search.c:1052-1484
 * One deliberate difference: P-matrix updates of an evaluation go out in one
 * pll_update_prob_matrices call per partition (count = number of invalid
 * branches) unless PLLHIP_EVAL_PMATRIX_PER_BRANCH is set, which reproduces the
 * reference's one-call-per-branch pattern (src/tree/treeinfo.c:845-865).
 */
#ifndef PLLHIP_EVAL_H_INCLUDED
#define PLLHIP_EVAL_H_INCLUDED

#include "pll.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PLLHIP_EVAL_PMATRIX_PER_BRANCH (1 << 0)
/* Newton-Raphson trial lengths.  The reference evaluates the current iterate per sumtable
   scan (src/optimize/pll_optimize.c:1223-1287).  Where a scan evaluates several lengths
   at the price of one (pllhip_free_trial_lengths() of EVERY partition of every worker,
   agreed through one MIN reduce) the driver also evaluates the iterates the step rule can
   produce by CLAMPING (+-dxmax, the bracket ends: known before the derivatives are), and
   an iteration whose point was already evaluated costs no scan.  The iterate sequence
   and every result are bit-identical either way.
   NO_SPECULATION: never; ALWAYS_SPECULATE: even where extra lengths cost time (tests). */
#define PLLHIP_EVAL_NO_SPECULATION     (1 << 1)
#define PLLHIP_EVAL_ALWAYS_SPECULATE   (1 << 2)

#define PLLHIP_EVAL_RADIUS_ALL (-1)   /* PLLMOD_OPT_BRLEN_OPTIMIZE_ALL, src/optimize/pll_optimize.h:102 */

/* error codes of the reference's optimiser that this driver reports
   (src/optimize/pll_optimize.h:88-99) */
#define PLLHIP_EVAL_ERROR_NEWTON_DERIV 2210
#define PLLHIP_EVAL_ERROR_NEWTON_LIMIT 2220
#define PLLHIP_EVAL_ERROR_NEWTON_WORSE 2240

/* branch-length linkage across partitions (PLLMOD_COMMON_BRLEN_*, src/pllmod_common.h:25-27):
   LINKED    one length per branch, the tree's;
   SCALED    the tree's length times a per-partition scaler (src/tree/treeinfo.c:849-852,
             chain rule in the derivatives: src/optimize/pll_optimize.c:1240-1262);
   UNLINKED  every partition has its own length per branch (src/tree/treeinfo.c:176-183);
             Newton-Raphson then runs one function per partition with its own bracket and
             convergence flag (src/optimize/opt_algorithms.c:133-261), the lengths of
             partitions other workers own arrive through a MAX reduce
             (src/optimize/pll_optimize.c:1431-1442). */
#define PLLHIP_EVAL_BRLEN_LINKED   0
#define PLLHIP_EVAL_BRLEN_SCALED   1
#define PLLHIP_EVAL_BRLEN_UNLINKED 2

typedef struct pllhip_eval pllhip_eval_t;

typedef void (*pllhip_reduce_fn)(void * ctx, double * data, size_t n, int op);

/* `tree` is borrowed (not destroyed); its records need unique node_index values
   (as produced by pll_utree_parse_newick* / pll_utree_reset_template_indices) */
PLL_EXPORT pllhip_eval_t * pllhip_eval_create(pll_utree_t * tree, unsigned int partition_count,
                                              unsigned int flags);
PLL_EXPORT void pllhip_eval_destroy(pllhip_eval_t * ev);

/* partition == NULL marks a partition that another worker owns */
PLL_EXPORT int pllhip_eval_set_partition(pllhip_eval_t * ev, unsigned int index,
                                         pll_partition_t * partition,
                                         const unsigned int * params_indices);

PLL_EXPORT void pllhip_eval_set_parallel_context(pllhip_eval_t * ev, void * ctx,
                                                 pllhip_reduce_fn reduce_cb);

/* Deferred scalar results (include/pllhip.h, pllhip_results_*): with such a table the
   driver enqueues the edge log-likelihoods / derivative scans of ALL its partitions,
   then fetches once -- one wait per evaluation or Newton-Raphson round instead of one per
   partition, and the sum over the workers happens on the device (RCCL) inside the fetch.
   It replaces the reduce callback for these two reductions.  The driver owns `results`
   and calls destroy() on it.  libpll_hip.so provides pllhip_eval_attach_comm() to fill
   this in; the table keeps this driver free of device code. */
typedef struct pllhip_eval_fused
{
  void * results;
  int (*edge_loglikelihood)(void * results, unsigned int slot, pll_partition_t * partition,
                            unsigned int parent_clv_index, int parent_scaler_index,
                            unsigned int child_clv_index, int child_scaler_index,
                            unsigned int matrix_index, const unsigned int * freqs_indices);
  int (*derivatives)(void * results, unsigned int slot, pll_partition_t * partition,
                     int parent_scaler_index, int child_scaler_index,
                     const double * branch_lengths, unsigned int count,
                     const unsigned int * params_indices, const double * sumtable);
  int (*fetch)(void * results, unsigned int first, unsigned int count, int op, double * out);
  void (*destroy)(void * results);
  /* a local failure before the fetch (this worker could not compute or enqueue its part): the next fetch
     contributes NaN and reports failure, so that all workers fail in the same call instead of one of them
     leaving the others inside the collective.  May be NULL. */
  void (*poison)(void * results);
} pllhip_eval_fused_t;

PLL_EXPORT void pllhip_eval_set_fused(pllhip_eval_t * ev, const pllhip_eval_fused_t * fused);

/* linkage mode (default LINKED).  UNLINKED copies the tree's current lengths into every
   partition's own array, like pllmod_treeinfo_create does (src/tree/treeinfo.c:1266-1277). */
PLL_EXPORT int pllhip_eval_set_brlen_linkage(pllhip_eval_t * ev, int linkage);
PLL_EXPORT int pllhip_eval_set_brlen_scaler(pllhip_eval_t * ev, unsigned int partition, double scaler);
PLL_EXPORT double pllhip_eval_get_brlen_scaler(const pllhip_eval_t * ev, unsigned int partition);
/* the length partition `partition` sees on `edge` before scaling: its own (UNLINKED) or the tree's */
PLL_EXPORT double pllhip_eval_get_partition_branch_length(const pllhip_eval_t * ev, unsigned int partition,
                                                          const pll_unode_t * edge);
/* UNLINKED only (src/tree/treeinfo.c:456-472) */
PLL_EXPORT int pllhip_eval_set_partition_branch_length(pllhip_eval_t * ev, unsigned int partition,
                                                       const pll_unode_t * edge, double length);

PLL_EXPORT int pllhip_eval_set_root(pllhip_eval_t * ev, pll_unode_t * root);
PLL_EXPORT pll_unode_t * pllhip_eval_root(const pllhip_eval_t * ev);

PLL_EXPORT void pllhip_eval_invalidate_all(pllhip_eval_t * ev);
PLL_EXPORT void pllhip_eval_invalidate_pmatrix(pllhip_eval_t * ev, const pll_unode_t * edge);
PLL_EXPORT void pllhip_eval_invalidate_clv(pllhip_eval_t * ev, const pll_unode_t * node);

/* log-likelihood at the current root edge; incremental != 0 recomputes only
   invalid P-matrices and CLVs.  NaN on error (pll_errno set). */
PLL_EXPORT double pllhip_eval_loglh(pllhip_eval_t * ev, int incremental);

/* Evaluate-only traversals (include/pllhip.h, pllhip_set_transient) for FULL evaluations -- the call the
   reference's model-parameter optimisers make after every parameter change
   (pllmod_treeinfo_compute_loglh(treeinfo, 0): src/algorithm/algo_callback.c:338, 465, 568, 678;
   nmax + 1 of them per L-BFGS-B iteration: src/optimize/opt_algorithms.c:734-773).  A full evaluation declares
   every vector invalid first, so what an earlier one did not store is given up (pllhip_discard_transient), and
   the vectors inside the operation chains of this one stay in registers.  Whatever reads a vector later
   (an incremental evaluation from another root, a branch-length optimisation, an SPR round) gets it recomputed on
   demand: results are bit-identical in every mode.
     OFF   never (default: pll_update_partials stores every vector, like the reference)
     ON    every full evaluation
     AUTO  a full evaluation that directly follows a full evaluation (the optimiser pattern) */
#define PLLHIP_EVAL_TRANSIENT_OFF  0
#define PLLHIP_EVAL_TRANSIENT_ON   1
#define PLLHIP_EVAL_TRANSIENT_AUTO 2
PLL_EXPORT void pllhip_eval_set_transient(pllhip_eval_t * ev, int mode);

/* change a branch length and invalidate what depends on it */
PLL_EXPORT void pllhip_eval_set_branch_length(pllhip_eval_t * ev, pll_unode_t * edge, double length);

/* returns the NEGATIVE log-likelihood after optimisation (like the reference);
   CLVs need not be valid on entry.  0 on error. */
PLL_EXPORT double pllhip_eval_optimize_branches(pllhip_eval_t * ev, double min_brlen, double max_brlen,
                                                double lh_epsilon, int max_iters, int radius);

/* work counters: operations and P-matrices handed to libpll so far */
/* ---------------------------------------------------------------------------
 * One SPR round: the counterpart of pllmod_algo_spr_round
 * (src/algorithm/algo_search.c:1052-1484, linked branch lengths, no topological
 * constraint, incremental CLV updates) on this driver, for machines where
 * pll-modules cannot go.  Same decision procedure, so the same moves on the same
 * data:
 *   scan   every inner record p, in the reference's post-order (algo_search.c:94-121),
 *          is pruned and scored at every branch within [radius_min, radius_max]
 *          of the pruning point (breadth-first, descent stopped by the lnL cutoff;
 *          algo_search.c:603-899).  A placement that beats the current best by
 *          more than 1e-6 is applied at once and remembered for undo; otherwise
 *          it competes for the list of the ntopol_keep (x3 in fast mode) best
 *          placements (algo_search.c:905-1050).
 *   rescore all branches are optimised (smoothings/4 passes), then every
 *          remembered topology -- applied moves undone one by one, listed
 *          placements re-applied -- gets the same optimisation; the best one
 *          (by more than 0.01 lnL units) is restored (algo_search.c:1233-1445).
 * The reference's second, "thorough" scan of the listed nodes in fast mode is
 * unreachable there (its guard reads a counter that is never advanced,
 * algo_search.c:1193) and is therefore not built.
 * Returns the final log-likelihood (checked against a full re-evaluation to 1e-6,
 * algo_search.c:1453-1457); 0 on error (pll_errno set).
 * ------------------------------------------------------------------------- */
typedef struct pllhip_spr_params
{
  unsigned int radius_min;          /* >= 1 */
  unsigned int radius_max;
  unsigned int ntopol_keep;
  int thorough;                     /* optimise the three branches at every insertion */
  double bl_min, bl_max;
  int smoothings;
  double epsilon;                   /* lnL epsilon of the whole-tree optimisations */
  double subtree_cutoff;            /* multiplier of the average lnL loss -> next round's cutoff */
  double lh_epsilon_brlen_triplet;
} pllhip_spr_params_t;

/* carried from round to round (cutoff_info_t, src/algorithm/pllmod_algorithm.h:41-47) */
typedef struct pllhip_spr_cutoff
{
  double lh_start;
  double lh_cutoff;
  double lh_dec_sum;
  int lh_dec_count;
} pllhip_spr_cutoff_t;

#define PLLHIP_SPR_LOG_MAX 256

typedef struct pllhip_spr_stats
{
  unsigned long prunings;           /* subtrees pruned and scanned */
  unsigned long insertions;         /* placements scored */
  unsigned long moves_applied;      /* improving moves applied during the scan */
  unsigned long rescored;           /* topologies that got a whole-tree optimisation */
  double lnl_start, lnl_scan, lnl_final;
  unsigned int log_count;           /* applied moves, in order: node_index of p and of r */
  unsigned int log_prune[PLLHIP_SPR_LOG_MAX];
  unsigned int log_regraft[PLLHIP_SPR_LOG_MAX];
} pllhip_spr_stats_t;

PLL_EXPORT double pllhip_eval_spr_round(pllhip_eval_t * ev, const pllhip_spr_params_t * params,
                                        pllhip_spr_cutoff_t * cutoff, pllhip_spr_stats_t * stats);

PLL_EXPORT unsigned long pllhip_eval_ops(const pllhip_eval_t * ev);
PLL_EXPORT unsigned long pllhip_eval_pmatrix_updates(const pllhip_eval_t * ev);
PLL_EXPORT unsigned long pllhip_eval_derivative_calls(const pllhip_eval_t * ev);   /* sumtable scans */
PLL_EXPORT unsigned long pllhip_eval_newton_iterations(const pllhip_eval_t * ev);  /* iterates consumed */

#ifdef __cplusplus
}
#endif

#endif /* PLLHIP_EVAL_H_INCLUDED */
