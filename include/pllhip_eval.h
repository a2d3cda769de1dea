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

#define PLLHIP_EVAL_RADIUS_ALL (-1)   /* PLLMOD_OPT_BRLEN_OPTIMIZE_ALL, src/optimize/pll_optimize.h:102 */

/* error codes of the reference's optimiser that this driver reports
   (src/optimize/pll_optimize.h:88-99) */
#define PLLHIP_EVAL_ERROR_NEWTON_DERIV 2210
#define PLLHIP_EVAL_ERROR_NEWTON_LIMIT 2220
#define PLLHIP_EVAL_ERROR_NEWTON_WORSE 2240

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

PLL_EXPORT int pllhip_eval_set_root(pllhip_eval_t * ev, pll_unode_t * root);
PLL_EXPORT pll_unode_t * pllhip_eval_root(const pllhip_eval_t * ev);

PLL_EXPORT void pllhip_eval_invalidate_all(pllhip_eval_t * ev);
PLL_EXPORT void pllhip_eval_invalidate_pmatrix(pllhip_eval_t * ev, const pll_unode_t * edge);
PLL_EXPORT void pllhip_eval_invalidate_clv(pllhip_eval_t * ev, const pll_unode_t * node);

/* log-likelihood at the current root edge; incremental != 0 recomputes only
   invalid P-matrices and CLVs.  NaN on error (pll_errno set). */
PLL_EXPORT double pllhip_eval_loglh(pllhip_eval_t * ev, int incremental);

/* change a branch length and invalidate what depends on it */
PLL_EXPORT void pllhip_eval_set_branch_length(pllhip_eval_t * ev, pll_unode_t * edge, double length);

/* returns the NEGATIVE log-likelihood after optimisation (like the reference);
   CLVs need not be valid on entry.  0 on error. */
PLL_EXPORT double pllhip_eval_optimize_branches(pllhip_eval_t * ev, double min_brlen, double max_brlen,
                                                double lh_epsilon, int max_iters, int radius);

/* work counters: operations and P-matrices handed to libpll so far */
PLL_EXPORT unsigned long pllhip_eval_ops(const pllhip_eval_t * ev);
PLL_EXPORT unsigned long pllhip_eval_pmatrix_updates(const pllhip_eval_t * ev);
PLL_EXPORT unsigned long pllhip_eval_derivative_calls(const pllhip_eval_t * ev);

#ifdef __cplusplus
}
#endif

#endif /* PLLHIP_EVAL_H_INCLUDED */
