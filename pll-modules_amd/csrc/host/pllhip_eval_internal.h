/*
 * pllhip_eval_internal.h -- shared between the evaluation driver
 * (pllhip_eval.c) and the topology search built on it (pllhip_search.c).
 * Not installed; the public interface is include/pllhip_eval.h.
 */
#ifndef PLLHIP_EVAL_INTERNAL_H_INCLUDED
#define PLLHIP_EVAL_INTERNAL_H_INCLUDED

#include "pllhip_eval.h"

struct pllhip_eval
{
  pll_utree_t * tree;
  pll_unode_t * root;
  unsigned int tips, inner, records, edges, nparts, flags;
  pll_partition_t ** parts;
  unsigned int ** params;         /* [partition][rate_cats] */
  double ** sumtables;            /* [partition], allocated on first use */
  char * clv_valid;               /* by node_index */
  char * pmat_valid;              /* by pmatrix_index */
  pll_unode_t ** trav;
  pll_operation_t * ops;
  double * brlens;
  unsigned int * midx;
  double * part_lnl;
  void * ctx;
  pllhip_reduce_fn reduce_cb;
  unsigned long n_ops, n_pmat, n_deriv, n_newton;
  pllhip_eval_fused_t fused;      /* fused.fetch != NULL: deferred results */
  int fused_auto;                 /* the result group was attached by pllhip_eval_create itself */
  int device_newton;              /* 1: try pllhip_newton_branch; 0: the library said it cannot (or PLLHIP_EVAL_DEVICE_NEWTON=0) */
  double * slot_buf;              /* [nparts * 2 * PLLHIP_EVAL_MAX_TRIALS] */
  unsigned int spec_trials;       /* trial lengths per scan; 0 = not decided yet */
  int linkage;                    /* PLLHIP_EVAL_BRLEN_* */
  double * brlen_scalers;         /* [nparts], 1.0 unless SCALED */
  double ** part_brlens;          /* [nparts][edges] by pmatrix_index, UNLINKED only (else NULL) */
  double * nr_x, * nr_xl, * nr_xh, * nr_f, * nr_df, * nr_orig;   /* [nparts] Newton-Raphson state, UNLINKED */
  int * nr_converged;
  int transient;                  /* PLLHIP_EVAL_TRANSIENT_*: which full evaluations run evaluate-only */
  int last_was_full;              /* the previous call into the driver was a full evaluation */
};

#define PLLHIP_EVAL_MAX_TRIALS 8
#define PLLHIP_EVAL_SPECULATE_MAX 4

void pllhip_eval_error(int code, const char * fmt, ...);
double pllhip_eval_optimize_impl(pllhip_eval_t * ev, double min_brlen, double max_brlen,
                                 double lh_epsilon, int max_iters, int radius, int keep_flags);

#endif
