/*
 * pllhip_eval.c -- evaluation driver on top of include/pll.h (see
 * include/pllhip_eval.h for scope and the reference functions it mirrors).
 * Plain C, no device code: everything below is calls into the libpll-style
 * interface, so the same file serves the HIP library and the CPU oracle.
 */
#include "pllhip_eval_internal.h"
#include "pllhip.h"
#pragma weak pllhip_eval_attach_comm   /* HIP engine only (pllhip_comm.hip) */
#pragma weak pllhip_newton_branch      /* HIP engine only (pll_core.hip) */
#pragma weak pllhip_newton_branch_multi
#pragma weak pllhip_set_transient      /* HIP engine only: the CPU oracle stores every vector */
#pragma weak pllhip_discard_transient
#include <stdarg.h>

static __thread pllhip_eval_t * cb_self;   /* pll_utree_traverse callbacks carry no user pointer */

void pllhip_eval_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, 200, fmt, ap);
  va_end(ap);
}

pllhip_eval_t * pllhip_eval_create(pll_utree_t * tree, unsigned int partition_count, unsigned int flags)
{
  unsigned int i;
  if (!tree || !tree->binary || !partition_count)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "pllhip_eval_create needs a binary tree and >= 1 partition");
    return NULL;
  }
  pllhip_eval_t * ev = (pllhip_eval_t *)calloc(1, sizeof(*ev));
  if (!ev) goto nomem;
  ev->tree = tree;
  ev->tips = tree->tip_count;
  ev->inner = tree->inner_count;
  ev->records = ev->tips + 3 * ev->inner;
  ev->edges = tree->edge_count;
  ev->nparts = partition_count;
  ev->flags = flags;
  ev->parts = (pll_partition_t **)calloc(partition_count, sizeof(*ev->parts));
  ev->params = (unsigned int **)calloc(partition_count, sizeof(*ev->params));
  ev->sumtables = (double **)calloc(partition_count, sizeof(*ev->sumtables));
  ev->part_lnl = (double *)calloc(partition_count, sizeof(double));
  ev->clv_valid = (char *)calloc(ev->records, 1);
  ev->pmat_valid = (char *)calloc(ev->edges, 1);
  ev->trav = (pll_unode_t **)calloc(ev->tips + ev->inner, sizeof(*ev->trav));
  ev->ops = (pll_operation_t *)calloc(ev->inner, sizeof(*ev->ops));
  ev->brlens = (double *)calloc((size_t)2 * ev->edges, sizeof(double));   /* tree lengths | one partition's view */
  ev->midx = (unsigned int *)calloc(ev->edges, sizeof(unsigned int));
  ev->slot_buf = (double *)calloc((size_t)partition_count * 2 * PLLHIP_EVAL_MAX_TRIALS, sizeof(double));
  ev->brlen_scalers = (double *)calloc(partition_count, sizeof(double));
  ev->nr_x = (double *)calloc((size_t)partition_count * 6, sizeof(double));
  ev->nr_converged = (int *)calloc(partition_count, sizeof(int));
  if (ev->nr_x)
  {
    ev->nr_xl = ev->nr_x + partition_count;      ev->nr_xh = ev->nr_xl + partition_count;
    ev->nr_f = ev->nr_xh + partition_count;      ev->nr_df = ev->nr_f + partition_count;
    ev->nr_orig = ev->nr_df + partition_count;
  }
  for (i = 0; ev->brlen_scalers && i < partition_count; ++i) ev->brlen_scalers[i] = 1.0;
  if (!ev->brlen_scalers || !ev->nr_x || !ev->nr_converged || !ev->slot_buf || !ev->parts || !ev->params || !ev->sumtables || !ev->part_lnl || !ev->clv_valid ||
      !ev->pmat_valid || !ev->trav || !ev->ops || !ev->brlens || !ev->midx)
    goto nomem;
  /* indices must address the flag arrays */
  for (i = 0; i < ev->tips + ev->inner; ++i)
  {
    pll_unode_t * n = tree->nodes[i], * s = n;
    do
    {
      if (s->node_index >= ev->records || s->pmatrix_index >= ev->edges)
      {
        pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "node/pmatrix index out of range in tree record");
        pllhip_eval_destroy(ev);
        return NULL;
      }
      s = s->next;
    } while (s && s != n);
  }
  ev->root = tree->vroot->next ? tree->vroot : tree->vroot->back;
  {
    const char * env = getenv("PLLHIP_EVAL_DEVICE_NEWTON");
    ev->device_newton = (pllhip_newton_branch && (!env || atoi(env))) ? 1 : 0;
    env = getenv("PLLHIP_EVAL_TRANSIENT");          /* 0 / 1 / 2: pllhip_eval_set_transient for every evaluator */
    if (env) ev->transient = atoi(env);
  }
  /* Several partitions: leave every partition's lnL / derivative totals on the device and wait ONCE per
     evaluation / Newton round (include/pllhip.h, pllhip_results_*) instead of once per partition -- the
     reference's per-partition loop (src/tree/treeinfo.c:1020-1056) with the waits taken out.  Only where the
     library has such result groups (the HIP engine; the CPU oracle does not export the symbol).
     A reduce callback set later replaces it (pllhip_eval_set_parallel_context).  PLLHIP_EVAL_DEFERRED=0: off. */
  if (partition_count > 1 && pllhip_eval_attach_comm)
  {
    const char * env = getenv("PLLHIP_EVAL_DEFERRED");
    if (!env || atoi(env))
    {
      if (pllhip_eval_attach_comm(ev, NULL)) ev->fused_auto = 1;
      else pll_errno = 0;
    }
  }
  return ev;
nomem:
  pllhip_eval_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate evaluator");
  pllhip_eval_destroy(ev);
  return NULL;
}

void pllhip_eval_destroy(pllhip_eval_t * ev)
{
  unsigned int p;
  if (!ev) return;
  for (p = 0; p < ev->nparts; ++p)
  {
    if (ev->params) free(ev->params[p]);
    if (ev->sumtables) free(ev->sumtables[p]);   /* pll_aligned_alloc memory is free()-able */
  }
  free(ev->parts); free(ev->params); free(ev->sumtables); free(ev->part_lnl);
  free(ev->clv_valid); free(ev->pmat_valid); free(ev->trav); free(ev->ops);
  free(ev->brlens); free(ev->midx); free(ev->slot_buf);
  free(ev->brlen_scalers); free(ev->nr_x); free(ev->nr_converged);
  if (ev->part_brlens)
  {
    for (p = 0; p < ev->nparts; ++p) free(ev->part_brlens[p]);
    free(ev->part_brlens);
  }
  if (ev->fused.destroy && ev->fused.results) ev->fused.destroy(ev->fused.results);
  free(ev);
}

int pllhip_eval_set_partition(pllhip_eval_t * ev, unsigned int index, pll_partition_t * partition,
                              const unsigned int * params_indices)
{
  unsigned int r;
  if (index >= ev->nparts)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "partition index %u out of range", index);
    return PLL_FAILURE;
  }
  if (partition)
  {
    if (partition->tips != ev->tips || partition->clv_buffers < ev->inner ||
        partition->prob_matrices < ev->edges)
    {
      pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "partition %u does not fit the tree", index);
      return PLL_FAILURE;
    }
    free(ev->params[index]);
    ev->params[index] = (unsigned int *)calloc(partition->rate_cats, sizeof(unsigned int));
    if (!ev->params[index])
    {
      pllhip_eval_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate params indices");
      return PLL_FAILURE;
    }
    for (r = 0; params_indices && r < partition->rate_cats; ++r) ev->params[index][r] = params_indices[r];
  }
  ev->parts[index] = partition;
  ev->spec_trials = 0;
  pllhip_eval_invalidate_all(ev);
  return PLL_SUCCESS;
}

void pllhip_eval_set_parallel_context(pllhip_eval_t * ev, void * ctx, pllhip_reduce_fn reduce_cb)
{
  /* sums go through the caller's hook from now on, not through the local result group of pllhip_eval_create */
  if (reduce_cb && ev->fused_auto) pllhip_eval_set_fused(ev, NULL);
  ev->ctx = ctx;
  ev->reduce_cb = reduce_cb;
  ev->spec_trials = 0;
}

void pllhip_eval_set_fused(pllhip_eval_t * ev, const pllhip_eval_fused_t * fused)
{
  if (ev->fused.destroy && ev->fused.results) ev->fused.destroy(ev->fused.results);
  memset(&ev->fused, 0, sizeof(ev->fused));
  ev->fused_auto = 0;
  if (fused) ev->fused = *fused;
}

int pllhip_eval_set_brlen_linkage(pllhip_eval_t * ev, int linkage)
{
  unsigned int p, i;
  if (linkage < PLLHIP_EVAL_BRLEN_LINKED || linkage > PLLHIP_EVAL_BRLEN_UNLINKED)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "unknown branch-length linkage %d", linkage);
    return PLL_FAILURE;
  }
  if (linkage == PLLHIP_EVAL_BRLEN_UNLINKED && !ev->part_brlens)
  {
    ev->part_brlens = (double **)calloc(ev->nparts, sizeof(double *));
    for (p = 0; ev->part_brlens && p < ev->nparts; ++p)
      if (!(ev->part_brlens[p] = (double *)calloc(ev->edges, sizeof(double)))) break;
    if (!ev->part_brlens || p < ev->nparts)
    {
      pllhip_eval_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate per-partition branch lengths");
      return PLL_FAILURE;
    }
    /* every partition starts from the tree's lengths */
    for (i = 0; i < ev->tips + ev->inner; ++i)
    {
      const pll_unode_t * n = ev->tree->nodes[i], * s2 = n;
      do
      {
        for (p = 0; p < ev->nparts; ++p) ev->part_brlens[p][s2->pmatrix_index] = s2->length;
        s2 = s2->next;
      } while (s2 && s2 != n);
    }
  }
  ev->linkage = linkage;
  if (linkage != PLLHIP_EVAL_BRLEN_SCALED)
    for (p = 0; p < ev->nparts; ++p) ev->brlen_scalers[p] = 1.0;
  pllhip_eval_invalidate_all(ev);
  return PLL_SUCCESS;
}

int pllhip_eval_set_brlen_scaler(pllhip_eval_t * ev, unsigned int partition, double scaler)
{
  if (partition >= ev->nparts || ev->linkage != PLLHIP_EVAL_BRLEN_SCALED || !(scaler > 0.0))
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "branch-length scalers need SCALED linkage, a valid "
                      "partition and a positive value");
    return PLL_FAILURE;
  }
  ev->brlen_scalers[partition] = scaler;
  pllhip_eval_invalidate_all(ev);
  return PLL_SUCCESS;
}

double pllhip_eval_get_brlen_scaler(const pllhip_eval_t * ev, unsigned int partition)
{
  return partition < ev->nparts ? ev->brlen_scalers[partition] : 0.0;
}

double pllhip_eval_get_partition_branch_length(const pllhip_eval_t * ev, unsigned int partition,
                                               const pll_unode_t * edge)
{
  if (partition >= ev->nparts) return 0.0;
  return ev->part_brlens ? ev->part_brlens[partition][edge->pmatrix_index] : edge->length;
}

int pllhip_eval_set_partition_branch_length(pllhip_eval_t * ev, unsigned int partition,
                                            const pll_unode_t * edge, double length)
{
  if (partition >= ev->nparts || ev->linkage != PLLHIP_EVAL_BRLEN_UNLINKED || !(length >= 0.0))
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "per-partition branch lengths need UNLINKED linkage");
    return PLL_FAILURE;
  }
  ev->part_brlens[partition][edge->pmatrix_index] = length;
  ev->pmat_valid[edge->pmatrix_index] = 0;
  memset(ev->clv_valid, 0, ev->records);
  return PLL_SUCCESS;
}

/* the branch length partition p's P-matrix of branch m is computed for */
static double effective_length(const pllhip_eval_t * ev, unsigned int p, unsigned int m, double tree_length)
{
  const double t = ev->part_brlens ? ev->part_brlens[p][m] : tree_length;
  return (ev->linkage == PLLHIP_EVAL_BRLEN_SCALED) ? t * ev->brlen_scalers[p] : t;
}

int pllhip_eval_set_root(pllhip_eval_t * ev, pll_unode_t * root)
{
  if (!root || !root->next)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "the root must be an inner node record");
    return PLL_FAILURE;
  }
  ev->root = root;
  return PLL_SUCCESS;
}

pll_unode_t * pllhip_eval_root(const pllhip_eval_t * ev) { return ev->root; }

void pllhip_eval_invalidate_all(pllhip_eval_t * ev)
{
  memset(ev->clv_valid, 0, ev->records);
  memset(ev->pmat_valid, 0, ev->edges);
}

void pllhip_eval_invalidate_pmatrix(pllhip_eval_t * ev, const pll_unode_t * edge)
{
  ev->pmat_valid[edge->pmatrix_index] = 0;
}

void pllhip_eval_invalidate_clv(pllhip_eval_t * ev, const pll_unode_t * node)
{
  ev->clv_valid[node->node_index] = 0;
}

void pllhip_eval_set_branch_length(pllhip_eval_t * ev, pll_unode_t * edge, double length)
{
  unsigned int p;
  edge->length = edge->back->length = length;
  for (p = 0; ev->part_brlens && p < ev->nparts; ++p) ev->part_brlens[p][edge->pmatrix_index] = length;
  ev->pmat_valid[edge->pmatrix_index] = 0;
  /* every CLV that looks across this edge depends on it: all records on the far
     side pointing this way.  Conservative and cheap: drop all CLV flags. */
  memset(ev->clv_valid, 0, ev->records);
}

/* the CLV slot of an inner node is shared by its three records */
static void mark_clv_valid(pllhip_eval_t * ev, const pll_unode_t * node)
{
  ev->clv_valid[node->node_index] = 1;
  ev->clv_valid[node->next->node_index] = 0;
  ev->clv_valid[node->next->next->node_index] = 0;
}

static int cb_all(pll_unode_t * node) { (void)node; return 1; }

static int cb_invalid_only(pll_unode_t * node)
{
  if (!node->next) return 0;
  return !cb_self->clv_valid[node->node_index];
}

static int update_pmatrices(pllhip_eval_t * ev)
{
  unsigned int n = 0, i, k = 0, p;
  if (!pll_utree_traverse(ev->root, PLL_TREE_TRAVERSE_POSTORDER, cb_all, ev->trav, &n)) return PLL_FAILURE;
  for (i = 0; i < n; ++i)
  {
    const pll_unode_t * node = ev->trav[i];
    if (ev->pmat_valid[node->pmatrix_index]) continue;
    ev->pmat_valid[node->pmatrix_index] = 1;
    ev->midx[k] = node->pmatrix_index;
    ev->brlens[k++] = node->length;
  }
  if (!k) return PLL_SUCCESS;
  for (p = 0; p < ev->nparts; ++p)
  {
    double * bl = ev->brlens;
    if (!ev->parts[p]) continue;
    if (ev->linkage != PLLHIP_EVAL_BRLEN_LINKED)
    {
      /* this partition's own view of the invalid branches (scaled or unlinked lengths) */
      bl = ev->brlens + ev->edges;
      for (i = 0; i < k; ++i) bl[i] = effective_length(ev, p, ev->midx[i], ev->brlens[i]);
    }
    if (ev->flags & PLLHIP_EVAL_PMATRIX_PER_BRANCH)
    {
      for (i = 0; i < k; ++i)
        if (!pll_update_prob_matrices(ev->parts[p], ev->params[p], &ev->midx[i], &bl[i], 1))
          return PLL_FAILURE;
    }
    else if (!pll_update_prob_matrices(ev->parts[p], ev->params[p], ev->midx, bl, k))
      return PLL_FAILURE;
  }
  ev->n_pmat += k;
  return PLL_SUCCESS;
}

/* A worker that fails locally must not leave its peers inside the reduction: it still takes part, with NaN,
   so that every worker fails in the same call (include/pllhip.h, pllhip_results_fetch).  `failed`: this
   worker could not bring its vectors up to date. */
static void poison_and_reduce(pllhip_eval_t * ev, double * buf, unsigned int n)
{
  unsigned int i;
  const int code = pll_errno;
  char msg[200];
  memcpy(msg, pll_errmsg, sizeof(msg));
  if (ev->fused.fetch)
  {
    if (ev->fused.poison) ev->fused.poison(ev->fused.results);
    (void)ev->fused.fetch(ev->fused.results, 0, n, 0 /* SUM */, buf);
  }
  else if (ev->reduce_cb)
  {
    for (i = 0; i < n; ++i) buf[i] = NAN;
    ev->reduce_cb(ev->ctx, buf, n, 0 /* SUM */);
  }
  for (i = 0; i < n; ++i) buf[i] = NAN;
  if (code) { pll_errno = code; memcpy(pll_errmsg, msg, sizeof(msg)); }    /* the cause, not the consequence */
}

static double edge_loglh(pllhip_eval_t * ev, const pll_unode_t * e, int failed)
{
  unsigned int p;
  double total = 0.0;
  if (failed)
  {
    poison_and_reduce(ev, ev->part_lnl, ev->nparts);
    return NAN;
  }
  if (ev->fused.fetch)
  {
    /* every partition's kernel is enqueued, then ONE fetch: all-reduce on the device, one wait */
    for (p = 0; p < ev->nparts; ++p)
      if (ev->parts[p] &&
          !ev->fused.edge_loglikelihood(ev->fused.results, p, ev->parts[p], e->clv_index, e->scaler_index,
                                        e->back->clv_index, e->back->scaler_index, e->pmatrix_index,
                                        ev->params[p]))
      {
        poison_and_reduce(ev, ev->part_lnl, ev->nparts);
        return NAN;
      }
    if (!ev->fused.fetch(ev->fused.results, 0, ev->nparts, 0 /* SUM */, ev->part_lnl)) return NAN;
  }
  else
  {
    for (p = 0; p < ev->nparts; ++p)
    {
      ev->part_lnl[p] = 0.0;
      if (!ev->parts[p]) continue;
      ev->part_lnl[p] = pll_compute_edge_loglikelihood(ev->parts[p], e->clv_index, e->scaler_index,
                                                       e->back->clv_index, e->back->scaler_index,
                                                       e->pmatrix_index, ev->params[p], NULL);
    }
    if (ev->reduce_cb) ev->reduce_cb(ev->ctx, ev->part_lnl, ev->nparts, 0 /* SUM */);
  }
  for (p = 0; p < ev->nparts; ++p) total += ev->part_lnl[p];
  return total;
}

/* PLLHIP_EVAL_FAULT=N (+ PLLHIP_EVAL_FAULT_RANK is up to the caller: set the variable on one worker only):
   the N-th evaluation of this process fails before its reduction -- for the tests of the path above */
static int injected_fault(void)
{
  static long at = -1, n = 0;
  if (at < 0) { const char * e = getenv("PLLHIP_EVAL_FAULT"); at = e ? atol(e) : 0; }
  if (at > 0 && ++n == at)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "injected evaluation failure");
    return 1;
  }
  return 0;
}

void pllhip_eval_set_transient(pllhip_eval_t * ev, int mode)
{
  ev->transient = mode;
  ev->last_was_full = 0;
}

/* mode of the partitions' engines for the operation list that follows */
static void partitions_transient(pllhip_eval_t * ev, int on, int discard)
{
  unsigned int p;
  if (!pllhip_set_transient || !pllhip_discard_transient) return;
  for (p = 0; p < ev->nparts; ++p)
  {
    if (!ev->parts[p]) continue;
    if (discard) (void)pllhip_discard_transient(ev->parts[p]);
    (void)pllhip_set_transient(ev->parts[p], on);
  }
}

double pllhip_eval_loglh(pllhip_eval_t * ev, int incremental)
{
  unsigned int n = 0, nops = 0, i;
  int failed = 0;
  /* a full evaluation recomputes every vector: what the last one kept in registers only is given up BEFORE the
     P-matrices change (nothing is recomputed for them), and this one may keep its own in registers */
  const int transient = !incremental && (ev->transient == PLLHIP_EVAL_TRANSIENT_ON ||
                                         (ev->transient == PLLHIP_EVAL_TRANSIENT_AUTO && ev->last_was_full));
  ev->last_was_full = !incremental;
  if (!incremental) pllhip_eval_invalidate_all(ev);
  pll_errno = 0;
  if (!incremental && ev->transient != PLLHIP_EVAL_TRANSIENT_OFF) partitions_transient(ev, transient, 1);
  failed = injected_fault() || !update_pmatrices(ev);

  cb_self = ev;
  if (!failed && !pll_utree_traverse(ev->root, PLL_TREE_TRAVERSE_POSTORDER, cb_invalid_only, ev->trav, &n))
    failed = 1;
  if (!failed) pll_utree_create_operations(ev->trav, n, NULL, NULL, ev->ops, NULL, &nops);
  if (!failed && nops)
  {
    /* every partition walks the same list (src/tree/treeinfo.c:1020-1056): one call, so that partitions
       of one kernel family can share their launches */
    if (!pllhip_update_partials_batch(ev->parts, ev->nparts, ev->ops, nops) || pll_errno) failed = 1;
    else
    {
      for (i = 0; i < n; ++i)
        if (ev->trav[i]->next) mark_clv_valid(ev, ev->trav[i]);
      ev->n_ops += nops;
    }
  }
  if (transient) partitions_transient(ev, 0, 0);
  return edge_loglh(ev, ev->root, failed);
}

/* ---------------------------------------------------------------------- */
/* Newton-Raphson branch-length optimisation, linked branch lengths        */
/* ---------------------------------------------------------------------- */

typedef struct
{
  pllhip_eval_t * ev;
  double bl_min, bl_max, tolerance;
  unsigned int max_newton;
} blo_t;

static int ensure_sumtables(pllhip_eval_t * ev)
{
  unsigned int p;
  for (p = 0; p < ev->nparts; ++p)
  {
    const pll_partition_t * part = ev->parts[p];
    if (!part || ev->sumtables[p]) continue;
    /* sized and aligned exactly as the reference does it (src/tree/treeinfo.c:333-340) */
    size_t len = (size_t)part->sites * part->rate_cats * part->states_padded;
    ev->sumtables[p] = (double *)pll_aligned_alloc((len ? len : 1) * sizeof(double), part->alignment);
    if (!ev->sumtables[p])
    {
      pllhip_eval_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate sumtable");
      return PLL_FAILURE;
    }
  }
  return PLL_SUCCESS;
}

/* first and second derivative of -lnL over all partitions at `count` trial branch
   lengths from ONE scan of every partition's sumtable (and one reduce over the workers).
   SCALED linkage: partition p is evaluated at s_p t and contributes s_p df, s_p^2 ddf (chain
   rule, src/optimize/pll_optimize.c:1240-1262). */
static int derivatives(pllhip_eval_t * ev, const pll_unode_t * e, const double * t, unsigned int count,
                       double * f, double * df)
{
  unsigned int p, k;
  double * buf = ev->slot_buf;
  double tp[PLLHIP_EVAL_MAX_TRIALS];
  for (k = 0; k < count; ++k) f[k] = df[k] = 0.0;
  if (ev->fused.fetch)
  {
    for (p = 0; p < ev->nparts; ++p)
    {
      if (!ev->parts[p]) continue;
      for (k = 0; k < count; ++k) tp[k] = ev->brlen_scalers[p] * t[k];
      if (!ev->fused.derivatives(ev->fused.results, p * 2 * count, ev->parts[p], e->scaler_index,
                                 e->back->scaler_index, tp, count, ev->params[p], ev->sumtables[p]))
      {
        poison_and_reduce(ev, buf, ev->nparts * 2 * count);
        return PLL_FAILURE;
      }
    }
    if (!ev->fused.fetch(ev->fused.results, 0, ev->nparts * 2 * count, 0 /* SUM */, buf)) return PLL_FAILURE;
    for (p = 0; p < ev->nparts; ++p)
    {
      const double sc = ev->brlen_scalers[p];
      for (k = 0; k < count; ++k)
      {
        f[k] += sc * buf[(p * count + k) * 2];
        df[k] += sc * sc * buf[(p * count + k) * 2 + 1];
      }
    }
  }
  else
  {
    double a[PLLHIP_EVAL_MAX_TRIALS], b[PLLHIP_EVAL_MAX_TRIALS];
    for (p = 0; p < ev->nparts; ++p)
    {
      const double sc = ev->brlen_scalers[p];
      int ok;
      if (!ev->parts[p]) continue;
      for (k = 0; k < count; ++k) tp[k] = sc * t[k];
      if (count == 1)
        ok = pll_compute_likelihood_derivatives(ev->parts[p], e->scaler_index, e->back->scaler_index, tp[0],
                                                ev->params[p], ev->sumtables[p], &a[0], &b[0]);
      else
        ok = pllhip_compute_likelihood_derivatives_multi(ev->parts[p], e->scaler_index,
                                                         e->back->scaler_index, tp, count, ev->params[p],
                                                         ev->sumtables[p], a, b);
      if (!ok)
      {
        poison_and_reduce(ev, buf, 2 * count);
        return PLL_FAILURE;
      }
      for (k = 0; k < count; ++k) { f[k] += sc * a[k]; df[k] += sc * sc * b[k]; }
    }
    if (ev->reduce_cb)
    {
      /* {df, ddf} of every trial length in one message (src/optimize/pll_optimize.c:1281-1284) */
      for (k = 0; k < count; ++k) { buf[2 * k] = f[k]; buf[2 * k + 1] = df[k]; }
      ev->reduce_cb(ev->ctx, buf, 2 * count, 0 /* SUM */);
      for (k = 0; k < count; ++k) { f[k] = buf[2 * k]; df[k] = buf[2 * k + 1]; }
    }
  }
  ev->n_deriv++;
  return PLL_SUCCESS;
}

/* UNLINKED: every partition at its own length x[p]; f[p], df[p] stay per partition
   (src/optimize/pll_optimize.c:1229-1279; one message of 2 P values here) */
static int derivatives_unlinked(pllhip_eval_t * ev, const pll_unode_t * e, const double * x,
                                double * f, double * df)
{
  unsigned int p;
  double * buf = ev->slot_buf;
  if (ev->fused.fetch)
  {
    for (p = 0; p < ev->nparts; ++p)
      if (ev->parts[p] &&
          !ev->fused.derivatives(ev->fused.results, 2 * p, ev->parts[p], e->scaler_index,
                                 e->back->scaler_index, &x[p], 1, ev->params[p], ev->sumtables[p]))
      {
        poison_and_reduce(ev, buf, 2 * ev->nparts);
        return PLL_FAILURE;
      }
    if (!ev->fused.fetch(ev->fused.results, 0, 2 * ev->nparts, 0 /* SUM */, buf)) return PLL_FAILURE;
  }
  else
  {
    for (p = 0; p < ev->nparts; ++p)
    {
      buf[2 * p] = buf[2 * p + 1] = 0.0;
      if (ev->parts[p] &&
          !pll_compute_likelihood_derivatives(ev->parts[p], e->scaler_index, e->back->scaler_index, x[p],
                                              ev->params[p], ev->sumtables[p], &buf[2 * p], &buf[2 * p + 1]))
      {
        poison_and_reduce(ev, buf, 2 * ev->nparts);
        return PLL_FAILURE;
      }
    }
    if (ev->reduce_cb) ev->reduce_cb(ev->ctx, buf, 2 * ev->nparts, 0 /* SUM */);
  }
  for (p = 0; p < ev->nparts; ++p) { f[p] = buf[2 * p]; df[p] = buf[2 * p + 1]; }
  ev->n_deriv++;
  return PLL_SUCCESS;
}

/* trial lengths per scan: every worker must scan with the same number (the reduce payloads
   have to match), so the local minimum over the partitions goes through one MIN reduce */
static unsigned int speculation_width(pllhip_eval_t * ev)
{
  unsigned int p;
  double v = PLLHIP_EVAL_SPECULATE_MAX;
  if (ev->spec_trials) return ev->spec_trials;
  if (ev->flags & PLLHIP_EVAL_NO_SPECULATION) v = 1;
  else if (!(ev->flags & PLLHIP_EVAL_ALWAYS_SPECULATE))
  {
    for (p = 0; p < ev->nparts; ++p)
      if (ev->parts[p]) v = PLL_MIN(v, (double)pllhip_free_trial_lengths(ev->parts[p]));
    if (ev->reduce_cb) ev->reduce_cb(ev->ctx, &v, 1, 2 /* MIN */);
    if (v < 3) v = 1;             /* the current iterate + both clamped outcomes, or nothing */
  }
  ev->spec_trials = (unsigned int)v;
  return ev->spec_trials;
}

/* the step rule of the reference's minimiser (opt_algorithms.c:208-240) clamps a Newton
   step to +-dxmax and to the bracket [xl, xh]: those outcomes are known BEFORE the
   derivatives are.  Returns the iterate a clamped step upwards (dir > 0) / downwards leads
   to from x, computed with the step rule's own expressions so that it is that iterate bit
   for bit. */
static double clamped_iterate(const blo_t * b, double x, double xl, double xh, double dxmax, int dir)
{
  double dx = dir > 0 ? dxmax : -dxmax;
  if (x + dx < xl) dx = xl - x;
  if (x + dx > xh) dx = xh - x;
  x += dx;
  return PLL_MAX(PLL_MIN(x, b->bl_max), b->bl_min);
}

static unsigned int add_trial(double * t, unsigned int n, double v)
{
  unsigned int i;
  for (i = 0; i < n; ++i) if (t[i] == v) return n;
  if (n < PLLHIP_EVAL_MAX_TRIALS) t[n++] = v;
  return n;
}

/* one-dimensional Newton-Raphson with bracketing: step rule of the reference's
   multi-function minimiser for a single function (opt_algorithms.c:133-261).  Each scan
   of the sumtables evaluates the current iterate and the iterates clamped steps would
   lead to (two levels); an iteration whose point is among the ones already evaluated
   costs no scan.  Same iterates, same results as one scan per iteration. */
static int newton(const blo_t * b, const pll_unode_t * e, double * x)
{
  pllhip_eval_t * ev = b->ev;
  const double dxmax = b->bl_max / b->max_newton;
  const unsigned int width = speculation_width(ev);     /* UNLINKED runs newton_unlinked instead */
  const int speculate = width > 1;
  double xl = b->bl_min, xh = b->bl_max, f, df, dx;
  double t[PLLHIP_EVAL_MAX_TRIALS], tf[PLLHIP_EVAL_MAX_TRIALS], tdf[PLLHIP_EVAL_MAX_TRIALS];
  unsigned int nt = 0, k, iter = 0;
  /* Every partition on this worker, one length per branch (linked, or scaled by a per-partition factor): the whole
     loop runs on the device (include/pllhip.h, pllhip_newton_branch / _multi: the same iterates, one wait per branch
     instead of one per iterate).  With other workers the sums of the derivatives have to meet on the host between
     two iterates: the loop below. */
  /* (a result group with a communicator installs the reduce callback too: pllhip_eval_attach_comm) */
  /* (PLLHIP_EVAL_ALWAYS_SPECULATE asks for the host loop's trial lengths: tests) */
  if (ev->device_newton && !ev->reduce_cb && !ev->part_brlens && !(ev->flags & PLLHIP_EVAL_ALWAYS_SPECULATE) &&
      (ev->nparts == 1 ? (ev->parts[0] && ev->brlen_scalers[0] == 1.0)
                       : (pllhip_newton_branch_multi != NULL && ev->nparts <= 8)))
  {
    unsigned int its = 0, p;
    double len = *x;
    int ok, local = 1;
    for (p = 0; p < ev->nparts; ++p) if (!ev->parts[p]) local = 0;
    if (!local) ok = 0, pll_errno = PLLHIP_ERROR_NEWTON_UNSUPPORTED;
    else if (ev->nparts == 1)
      ok = pllhip_newton_branch(ev->parts[0], e->scaler_index, e->back->scaler_index, ev->params[0], ev->sumtables[0],
                                *x, b->bl_min, b->bl_max, b->tolerance, b->max_newton, &len, &its, NULL);
    else
      ok = pllhip_newton_branch_multi(ev->parts, ev->nparts, e->scaler_index, e->back->scaler_index,
                                      (const unsigned int * const *)ev->params, (const double * const *)ev->sumtables,
                                      ev->brlen_scalers, *x, b->bl_min, b->bl_max, b->tolerance, b->max_newton,
                                      &len, &its, NULL);
    if (ok)
    {
      *x = len;
      ev->n_newton += its;
      ev->n_deriv += its;
      return PLL_SUCCESS;
    }
    if (pll_errno == PLLHIP_ERROR_NEWTON_LIMIT)
    {
      /* (counted like the loop below: it gives up BEFORE the evaluation that would exceed the limit, the device
         loop after it) */
      if (its) { ev->n_newton += its - 1; ev->n_deriv += its - 1; }
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_LIMIT, "Exceeded maximum number of iterations");
      return PLL_FAILURE;
    }
    if (pll_errno == PLLHIP_ERROR_NEWTON_DERIVATIVES)
    {
      ev->n_newton += its;
      ev->n_deriv += its;
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_DERIV, "Wrong likelihood derivatives");
      return PLL_FAILURE;
    }
    /* UNSUPPORTED: this partition cannot; STUCK: the device is shared and the loop's workgroups did not all get
       onto it (nothing was changed): either way this branch -- and the ones after it -- iterate on the host */
    if (pll_errno != PLLHIP_ERROR_NEWTON_UNSUPPORTED && pll_errno != PLLHIP_ERROR_NEWTON_STUCK) return PLL_FAILURE;
    pll_errno = 0;
    ev->device_newton = 0;
  }
  *x = PLL_MAX(PLL_MIN(*x, b->bl_max), b->bl_min);
  for (;;)
  {
    if (iter++ > b->max_newton)
    {
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_LIMIT, "Exceeded maximum number of iterations");
      return PLL_FAILURE;
    }
    for (k = 0; k < nt; ++k) if (t[k] == *x) break;
    if (k == nt)
    {
      nt = 0;
      t[nt++] = *x;
      if (speculate)
      {
        const double up = clamped_iterate(b, *x, xl, xh, dxmax, 1);
        const double dn = clamped_iterate(b, *x, xl, xh, dxmax, -1);
        nt = add_trial(t, nt, dn);
        nt = add_trial(t, nt, up);
        /* second level, while the scan stays at four lengths (beyond that the extra
           arithmetic of a length shows in the scan time): brackets as they stand after a
           step in that direction */
        if (nt < width) nt = add_trial(t, nt, clamped_iterate(b, dn, xl, *x, dxmax, -1));
        if (nt < width) nt = add_trial(t, nt, clamped_iterate(b, up, *x, xh, dxmax, 1));
      }
      if (!derivatives(ev, e, t, nt, tf, tdf)) return PLL_FAILURE;
      k = 0;
    }
    f = tf[k];
    df = tdf[k];
    ev->n_newton++;
    if (!isfinite(f) || !isfinite(df))
    {
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_DERIV, "Wrong likelihood derivatives");
      return PLL_FAILURE;
    }
    if (df > 0.0)
    {
      if (fabs(f) < b->tolerance) return PLL_SUCCESS;
      if (f < 0.0) xl = *x; else xh = *x;
      dx = -f / df;
    }
    else
      dx = -f / fabs(df);
    dx = PLL_MAX(PLL_MIN(dx, dxmax), -dxmax);
    if (*x + dx < xl) dx = xl - *x;
    if (*x + dx > xh) dx = xh - *x;
    if (fabs(dx) < b->tolerance) return PLL_SUCCESS;
    *x += dx;
    *x = PLL_MAX(PLL_MIN(*x, b->bl_max), b->bl_min);
  }
}

/* UNLINKED: one function per partition, each with its own bracket and convergence flag,
   all evaluated by one scan per iteration (pllmod_opt_minimize_newton_multi,
   src/optimize/opt_algorithms.c:133-261).  ev->nr_x holds the iterates. */
static int newton_unlinked(const blo_t * b, const pll_unode_t * e)
{
  pllhip_eval_t * ev = b->ev;
  const unsigned int n = ev->nparts;
  const double dxmax = b->bl_max / b->max_newton;
  double * x = ev->nr_x, * xl = ev->nr_xl, * xh = ev->nr_xh, * f = ev->nr_f, * df = ev->nr_df;
  int * converged = ev->nr_converged;
  unsigned int i, iter = 0;
  int all_converged = 0;
  for (i = 0; i < n; ++i)
  {
    x[i] = PLL_MAX(PLL_MIN(x[i], b->bl_max), b->bl_min);
    xl[i] = b->bl_min;
    xh[i] = b->bl_max;
    converged[i] = 0;
  }
  while (!all_converged)
  {
    if (iter++ > b->max_newton)
    {
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_LIMIT, "Exceeded maximum number of iterations");
      return PLL_FAILURE;
    }
    if (!derivatives_unlinked(ev, e, x, f, df)) return PLL_FAILURE;
    ev->n_newton++;
    all_converged = 1;
    for (i = 0; i < n; ++i)
    {
      double dx;
      if (converged[i]) continue;
      if (!isfinite(f[i]) || !isfinite(df[i]))
      {
        pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_DERIV, "Wrong likelihood derivatives");
        return PLL_FAILURE;
      }
      if (df[i] > 0.0)
      {
        if (fabs(f[i]) < b->tolerance) { converged[i] = 1; continue; }
        if (f[i] < 0.0) xl[i] = x[i]; else xh[i] = x[i];
        dx = -f[i] / df[i];
      }
      else
        dx = -f[i] / fabs(df[i]);
      dx = PLL_MAX(PLL_MIN(dx, dxmax), -dxmax);
      if (x[i] + dx < xl[i]) dx = xl[i] - x[i];
      if (x[i] + dx > xh[i]) dx = xh[i] - x[i];
      if (fabs(dx) < b->tolerance) { converged[i] = 1; continue; }
      x[i] += dx;
      x[i] = PLL_MAX(PLL_MIN(x[i], b->bl_max), b->bl_min);
      all_converged = 0;
    }
  }
  return PLL_SUCCESS;
}

/* recompute the CLV at `parent` from the far ends of c1 and c2
   (update_partials_and_scalers, src/optimize/pll_optimize.c:748-775) */
static int reorient(pllhip_eval_t * ev, const pll_unode_t * parent, const pll_unode_t * c1,
                    const pll_unode_t * c2)
{
  pll_operation_t op;
  unsigned int p;
  op.parent_clv_index = parent->clv_index;
  op.parent_scaler_index = parent->scaler_index;
  op.child1_clv_index = c1->back->clv_index;
  op.child1_matrix_index = c1->back->pmatrix_index;
  op.child1_scaler_index = c1->back->scaler_index;
  op.child2_clv_index = c2->back->clv_index;
  op.child2_matrix_index = c2->back->pmatrix_index;
  op.child2_scaler_index = c2->back->scaler_index;
  (void)p;
  if (!pllhip_update_partials_batch(ev->parts, ev->nparts, &op, 1)) return PLL_FAILURE;
  ev->n_ops++;
  return pll_errno ? PLL_FAILURE : PLL_SUCCESS;
}

/* this branch's P-matrix in every local partition, at the lengths now in force */
static int refresh_pmatrix(pllhip_eval_t * ev, const pll_unode_t * edge)
{
  unsigned int p;
  for (p = 0; p < ev->nparts; ++p)
  {
    double t;
    if (!ev->parts[p]) continue;
    t = effective_length(ev, p, edge->pmatrix_index, edge->length);
    if (!pll_update_prob_matrices(ev->parts[p], ev->params[p], &edge->pmatrix_index, &t, 1))
      return PLL_FAILURE;
  }
  ev->n_pmat++;
  return PLL_SUCCESS;
}

static int optimise_around(const blo_t * b, pll_unode_t * p_edge, int radius)
{
  pllhip_eval_t * ev = b->ev;
  pll_unode_t * q = p_edge->next, * z = q ? q->next : NULL;
  const unsigned int m = p_edge->pmatrix_index;
  unsigned int p;
  int changed = 0;

  for (p = 0; p < ev->nparts; ++p)
    if (ev->parts[p] &&
        !pll_update_sumtable(ev->parts[p], p_edge->clv_index, p_edge->back->clv_index,
                             p_edge->scaler_index, p_edge->back->scaler_index, ev->params[p],
                             ev->sumtables[p]))
      return PLL_FAILURE;

  if (ev->linkage == PLLHIP_EVAL_BRLEN_UNLINKED)
  {
    double * x = ev->nr_x, * orig = ev->nr_orig;
    for (p = 0; p < ev->nparts; ++p) x[p] = ev->parts[p] ? ev->part_brlens[p][m] : 0.0;
    /* every worker needs all per-partition lengths (src/optimize/pll_optimize.c:1436-1442) */
    if (ev->nparts > 1 && ev->reduce_cb) ev->reduce_cb(ev->ctx, x, ev->nparts, 1 /* MAX */);
    memcpy(orig, x, sizeof(double) * ev->nparts);
    if (!newton_unlinked(b, p_edge))
    {
      if (pll_errno != PLLHIP_EVAL_ERROR_NEWTON_LIMIT) return PLL_FAILURE;
      for (p = 0; p < ev->nparts; ++p)
        if (!ev->nr_converged[p]) x[p] = orig[p];      /* no lnL check: keep the old length */
      pll_errno = 0;
    }
    for (p = 0; p < ev->nparts; ++p)
    {
      if (fabs(x[p] - orig[p]) < 1e-10) continue;
      ev->part_brlens[p][m] = x[p];
      changed = 1;
    }
  }
  else
  {
    const double xorig = p_edge->length;
    double x = xorig;
    if (!newton(b, p_edge, &x))
    {
      if (pll_errno != PLLHIP_EVAL_ERROR_NEWTON_LIMIT) return PLL_FAILURE;
      x = xorig;                    /* no convergence, no lnL check: keep the old length */
      pll_errno = 0;
    }
    if (fabs(x - xorig) >= 1e-10)
    {
      p_edge->length = p_edge->back->length = x;
      changed = 1;
    }
  }
  if (changed && !refresh_pmatrix(ev, p_edge)) return PLL_FAILURE;
  if (radius && q && z)
  {
    if (!reorient(ev, q, p_edge, z) || !optimise_around(b, q->back, radius - 1)) return PLL_FAILURE;
    if (!reorient(ev, z, q, p_edge) || !optimise_around(b, z->back, radius - 1)) return PLL_FAILURE;
    if (!reorient(ev, p_edge, z, q)) return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

double pllhip_eval_optimize_branches(pllhip_eval_t * ev, double min_brlen, double max_brlen,
                                     double lh_epsilon, int max_iters, int radius)
{
  return pllhip_eval_optimize_impl(ev, min_brlen, max_brlen, lh_epsilon, max_iters, radius, 0);
}

/* keep_flags: the caller tracks validity itself (the SPR scan optimises the three
   branches around an insertion point and restores them afterwards) */
double pllhip_eval_optimize_impl(pllhip_eval_t * ev, double min_brlen, double max_brlen,
                                 double lh_epsilon, int max_iters, int radius, int keep_flags)
{
  blo_t b;
  double lnl, new_lnl;
  int iters = max_iters;
  pll_unode_t * root = ev->root;
  if (radius < PLLHIP_EVAL_RADIUS_ALL)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "Invalid radius for branch length optimization");
    return 0.0;
  }
  b.ev = ev;
  b.bl_min = (min_brlen > 0) ? min_brlen : 1e-6;
  b.bl_max = (max_brlen > 0) ? max_brlen : 100.0;
  b.tolerance = (min_brlen > 0) ? min_brlen / 10.0 : 1e-4;
  b.max_newton = 30;
  if (!ensure_sumtables(ev)) return 0.0;

  /* precondition of the reference: CLVs valid towards the root, P-matrices current */
  lnl = pllhip_eval_loglh(ev, 1);
  if (isnan(lnl)) return 0.0;

  while (iters)
  {
    if (!optimise_around(&b, root, radius)) return 0.0;
    if (radius && !optimise_around(&b, root->back, radius - 1)) return 0.0;
    new_lnl = edge_loglh(ev, root->back, 0);
    if (new_lnl - lnl > new_lnl * 1e-13)
    {
      iters--;
      if (fabs(new_lnl - lnl) < lh_epsilon) iters = 0;
      lnl = new_lnl;
    }
    else
    {
      pllhip_eval_error(PLLHIP_EVAL_ERROR_NEWTON_WORSE,
                 "BL opt converged to a worse likelihood score by %.15f units", new_lnl - lnl);
      lnl = new_lnl;
      break;
    }
  }
  /* P-matrices are current for every branch; CLVs were last refreshed at
     different moments of the sweep: let the next evaluation rebuild them */
  if (!keep_flags) memset(ev->clv_valid, 0, ev->records);
  return -lnl;
}

unsigned long pllhip_eval_ops(const pllhip_eval_t * ev) { return ev->n_ops; }
unsigned long pllhip_eval_pmatrix_updates(const pllhip_eval_t * ev) { return ev->n_pmat; }
unsigned long pllhip_eval_derivative_calls(const pllhip_eval_t * ev) { return ev->n_deriv; }
unsigned long pllhip_eval_newton_iterations(const pllhip_eval_t * ev) { return ev->n_newton; }
