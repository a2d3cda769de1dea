/*
 * pllhip_search.c -- one SPR round on the evaluation driver (include/pllhip_eval.h).
 *
 * The reference's pllmod_algo_spr_round (src/algorithm/algo_search.c:1052-1484)
 * cannot travel to the GPU box; this file is its counterpart on pllhip_eval_t:
 * written from the behaviour of that function (scan order, acceptance rules,
 * thresholds, the order in which remembered topologies are re-scored), with its
 * own data structures.  Plain C on include/pll.h, so it serves the HIP library
 * and the CPU oracle alike; tests/test_dropin_modules.py runs the reference's
 * function and this one on the same data and compares the resulting trees.
 *
 * Scope: linked, scaled and unlinked branch lengths, no topological constraint, incremental CLV
 * updates (the reference's fast_clv_updates = 1).  With unlinked lengths every length the round saves,
 * joins, halves, clamps or restores is a vector -- the tree's own length plus one per partition, kept by
 * P-matrix index (src/algorithm/algo_search.c:399-565: algo_unode_fix_length, algo_utree_prune,
 * algo_utree_regraft, and the pllmod_treeinfo_get / set_branch_length_all calls of best_reinsert_edge,
 * src/algorithm/algo_search.c:639-641, 756, 811-813).
 *
 * Why the validity flags stay truthful through the scan: every scan starts with
 * an evaluation rooted at the pruned record, after which every CLV slot of the
 * tree points at the pruning point; a record that is valid afterwards was
 * computed looking at the then-root, so its region never contains the pruned
 * subtree, wherever that subtree is attached at the time.
 */
#include "pllhip_eval_internal.h"
#include <stdio.h>

/* PLLHIP_SPR_TRACE=1: one line per scan / applied move / re-scored topology on stderr */
static int trace_on(void)
{
  static int on = -1;
  if (on < 0) { const char * e = getenv("PLLHIP_SPR_TRACE"); on = (e && atoi(e)) ? 1 : 0; }
  return on;
}
#define TRACE(...) do { if (trace_on()) fprintf(stderr, "[spr] " __VA_ARGS__); } while (0)

/* a branch length as the round handles it: v[0] the tree's, v[1 + p] partition p's own (unlinked only);
   LV(s) values */
#define LV(s) ((s)->nl)

typedef struct
{
  pll_unode_t * p, * r;
  double * b1, * b2, * b3;     /* [LV] each */
  double lh;
  unsigned int undo_pos;       /* number of moves applied before this placement was scored */
} placement_t;

typedef struct
{
  pll_unode_t * p, * home;     /* home: p->next->back before the move */
  double * p_len, * left_len, * right_len, * regraft_len;     /* [LV] each */
} undo_t;

typedef struct
{
  unsigned int left, right, pmatrix_index;
  double * length;             /* [LV] */
} saved_edge_t;

typedef struct
{
  pllhip_eval_t * ev;
  const pllhip_spr_params_t * prm;
  pllhip_spr_cutoff_t * cut;
  pllhip_spr_stats_t * st;
  int thorough;
  unsigned int nl;             /* 1, or 1 + partitions with unlinked branch lengths */
  undo_t * ring;               /* applied moves, newest at cur - 1 (wraps) */
  size_t ring_size, ring_cur;
  unsigned int ring_round;
  placement_t * best;          /* sorted by lh, descending; p == NULL marks the end */
  size_t best_size;
  pll_unode_t ** records;      /* by node_index */
  pll_unode_t ** queue;        /* breadth-first regraft candidates */
  unsigned int * qdist;
  double * pool;               /* the length vectors of all of the above + scratch */
  double * tmp;                /* [8][LV] scratch vectors */
} search_t;

#define TMP(s, k) ((s)->tmp + (size_t)(k) * LV(s))

/* ---- graph surgery; index rule of the reference: the second record of a new
   edge takes the P-matrix index of the first (utree_operations.c:359-374) ---- */

static void connect(pll_unode_t * a, pll_unode_t * b, double length)
{
  a->back = b;
  b->back = a;
  a->length = b->length = length;
  b->pmatrix_index = a->pmatrix_index;
}

/* the length vector of an edge / an edge takes a length vector */
static void get_lens(const search_t * s, const pll_unode_t * e, double * out)
{
  unsigned int p;
  out[0] = e->length;
  for (p = 0; p + 1 < LV(s); ++p) out[1 + p] = s->ev->part_brlens[p][e->pmatrix_index];
}

static void set_lens(search_t * s, pll_unode_t * e, const double * in)
{
  unsigned int p;
  e->length = e->back->length = in[0];
  for (p = 0; p + 1 < LV(s); ++p) s->ev->part_brlens[p][e->pmatrix_index] = in[1 + p];
}

static void copy_lens(const search_t * s, double * dst, const double * src) { memcpy(dst, src, sizeof(double) * LV(s)); }

/* utree_operations.c:184-207 / algo_search.c:495-524: neighbours joined, lengths added (every partition's own
   as well); returns the joined edge */
static pll_unode_t * detach(search_t * s, pll_unode_t * p)
{
  pll_unode_t * u = p->next->back, * v = p->next->next->back;
  double * a = TMP(s, 6), * b = TMP(s, 7);
  unsigned int k;
  get_lens(s, u, a);
  get_lens(s, v, b);
  for (k = 0; k < LV(s); ++k) a[k] += b[k];
  connect(u, v, a[0]);
  set_lens(s, u, a);
  p->next->back = p->next->next->back = NULL;
  return u;
}

/* utree_operations.c:229-257 / algo_search.c:527-565: the target edge is split in half */
static void attach(search_t * s, pll_unode_t * p, pll_unode_t * r)
{
  pll_unode_t * r2 = r->back;
  double * a = TMP(s, 6);
  unsigned int k;
  get_lens(s, r, a);
  for (k = 0; k < LV(s); ++k) a[k] = a[k] / 2;
  connect(r, p->next, a[0]);
  connect(p->next->next, r2, a[0]);
  set_lens(s, r, a);
  set_lens(s, r2, a);
}

/* algo_search.c:421-459 (algo_unode_fix_length): every length of the edge into [bl_min, bl_max] */
static void clamp_len(search_t * s, pll_unode_t * e)
{
  unsigned int p;
  int changed = 0;
  if (LV(s) == 1)
  {
    if (e->length < s->prm->bl_min) { e->length = e->back->length = s->prm->bl_min; changed = 1; }
    else if (e->length > s->prm->bl_max) { e->length = e->back->length = s->prm->bl_max; changed = 1; }
  }
  else
    for (p = 0; p + 1 < LV(s); ++p)
    {
      double * len = &s->ev->part_brlens[p][e->pmatrix_index];
      if (*len < s->prm->bl_min) { *len = s->prm->bl_min; changed = 1; }
      else if (*len > s->prm->bl_max) { *len = s->prm->bl_max; changed = 1; }
    }
  if (changed) pllhip_eval_invalidate_pmatrix(s->ev, e);
}

static void root_at(search_t * s, pll_unode_t * n) { s->ev->root = n->next ? n : n->back; }

static void invalidate_triplet(search_t * s, pll_unode_t * p)
{
  pllhip_eval_invalidate_pmatrix(s->ev, p);
  pllhip_eval_invalidate_pmatrix(s->ev, p->next);
  pllhip_eval_invalidate_pmatrix(s->ev, p->next->next);
}

/* ---- lists ---- */

static size_t ring_pos(const search_t * s) { return s->ring_size * s->ring_round + s->ring_cur; }

static undo_t * ring_next(search_t * s)
{
  if (s->ring_cur + 1 < s->ring_size) s->ring_cur++;
  else { s->ring_round++; s->ring_cur = 0; }
  return s->ring + s->ring_cur;
}

static undo_t * ring_prev(search_t * s)
{
  if (s->ring_cur > 0) s->ring_cur--;
  else if (s->ring_round > 0 && s->ring_size > 0) { s->ring_round--; s->ring_cur = s->ring_size - 1; }
  else return NULL;
  return s->ring + s->ring_cur;
}

/* the list owns its length vectors: an insertion rotates the vectors of the entry that drops out to the new slot */
static void best_save(search_t * s, const placement_t * e)
{
  size_t i = 0, j;
  placement_t spare;
  while (i < s->best_size && s->best[i].p && e->lh < s->best[i].lh) ++i;
  if (i >= s->best_size) return;
  spare = s->best[s->best_size - 1];
  for (j = s->best_size - 1; j > i; --j) s->best[j] = s->best[j - 1];
  s->best[i].p = e->p; s->best[i].r = e->r; s->best[i].lh = e->lh; s->best[i].undo_pos = e->undo_pos;
  s->best[i].b1 = spare.b1; s->best[i].b2 = spare.b2; s->best[i].b3 = spare.b3;
  copy_lens(s, s->best[i].b1, e->b1);
  copy_lens(s, s->best[i].b2, e->b2);
  copy_lens(s, s->best[i].b3, e->b3);
}

static int best_next(const search_t * s, size_t pos, int i)
{
  do
  {
    ++i;
    if (i >= (int)s->best_size || !s->best[i].p) return -1;
  } while (s->best[i].undo_pos != pos);
  return i;
}

/* ---- whole-tree helpers ---- */

/* post-order over inner nodes, three records each (algo_search.c:94-121) */
static void list_records(pll_unode_t * n, pll_unode_t ** out, unsigned int * k)
{
  if (!n->next) return;
  list_records(n->next->back, out, k);
  list_records(n->next->next->back, out, k);
  out[(*k)++] = n->next->next;
  out[(*k)++] = n->next;
  out[(*k)++] = n;
}

static void collect_at_depth(pll_unode_t * n, unsigned int depth, unsigned int want, search_t * s,
                             unsigned int * k)
{
  if (depth == want) { s->queue[*k] = n; s->qdist[*k] = want; ++*k; }
  if (depth >= want || !n->next) return;
  collect_at_depth(n->next->back, depth + 1, want, s, k);
  collect_at_depth(n->next->next->back, depth + 1, want, s, k);
}

static void save_topology(const search_t * s, saved_edge_t * out, unsigned int * root_index)
{
  unsigned int i, k = 0;
  for (i = 0; i < s->ev->records; ++i)
  {
    const pll_unode_t * n = s->records[i];
    if (n->node_index < n->back->node_index)
    {
      out[k].left = n->node_index;
      out[k].right = n->back->node_index;
      out[k].pmatrix_index = n->pmatrix_index;
      get_lens(s, n, out[k].length);
      ++k;
    }
  }
  if (root_index) *root_index = s->ev->root->node_index;
}

static void load_topology(search_t * s, const saved_edge_t * in, unsigned int root_index)
{
  unsigned int k;
  for (k = 0; k < s->ev->edges; ++k)
  {
    pll_unode_t * a = s->records[in[k].left], * b = s->records[in[k].right];
    connect(a, b, in[k].length[0]);
    a->pmatrix_index = b->pmatrix_index = in[k].pmatrix_index;
    set_lens(s, a, in[k].length);
  }
  s->ev->root = s->records[root_index];
  pllhip_eval_invalidate_all(s->ev);
}

/* full evaluation, then `passes` optimisation sweeps over every branch */
static double optimise_all(search_t * s, double lh_epsilon, double smooth_factor)
{
  const int passes = (int)round(smooth_factor * s->prm->smoothings);
  double v;
  if (isnan(pllhip_eval_loglh(s->ev, 0))) return 0.0;
  v = pllhip_eval_optimize_impl(s->ev, s->prm->bl_min, s->prm->bl_max, lh_epsilon, passes,
                                PLLHIP_EVAL_RADIUS_ALL, 0);
  return v ? -v : 0.0;
}

/* ---- the scan of one pruned subtree (algo_search.c:603-899) ---- */

static int scan_placements(search_t * s, placement_t * entry)
{
  pllhip_eval_t * ev = s->ev;
  pll_unode_t * p = entry->p, * home, * r;
  double * z1 = TMP(s, 0), * z2 = TMP(s, 1), * z3 = TMP(s, 2);
  double * b1 = TMP(s, 3), * b2 = TMP(s, 4), * b3 = TMP(s, 5);
  unsigned int count = 0, j;
  double lh;

  get_lens(s, p, z1); get_lens(s, p->next, z2); get_lens(s, p->next->next, z3);
  entry->r = NULL;
  entry->lh = -INFINITY;

  /* every CLV slot looks at the pruning point before the subtree leaves it */
  root_at(s, p);
  pllhip_eval_invalidate_clv(ev, p);
  if (isnan(pllhip_eval_loglh(ev, 1))) return PLL_FAILURE;

  home = detach(s, p);
  clamp_len(s, home);
  root_at(s, home);
  pllhip_eval_invalidate_clv(ev, home);
  pllhip_eval_invalidate_clv(ev, home->back);
  pllhip_eval_invalidate_pmatrix(ev, home);

  memset(s->queue, 0, ev->edges * 2 * sizeof(*s->queue));
  collect_at_depth(ev->root, 0, s->prm->radius_min, s, &count);
  if (ev->root->back->next) collect_at_depth(ev->root->back, 0, s->prm->radius_min, s, &count);
  if (s->st) s->st->prunings++;

  for (j = 0; (r = s->queue[j]) != NULL; ++j)
  {
    double * regraft_len = s->tmp + (size_t)8 * LV(s);
    int descend;
    if (r == home || r == home->back) continue;

    get_lens(s, r, regraft_len);
    attach(s, p, r);
    root_at(s, p);
    pllhip_eval_invalidate_clv(ev, p);
    get_lens(s, p, b1); get_lens(s, p->next, b2); get_lens(s, p->next->next, b3);
    clamp_len(s, p->next);
    clamp_len(s, p->next->next);
    pllhip_eval_invalidate_pmatrix(ev, p->next);
    pllhip_eval_invalidate_pmatrix(ev, p->next->next);

    lh = pllhip_eval_loglh(ev, 1);
    if (isnan(lh)) return PLL_FAILURE;
    if (s->thorough)
    {
      const double v = pllhip_eval_optimize_impl(ev, s->prm->bl_min, s->prm->bl_max,
                                                 s->prm->lh_epsilon_brlen_triplet,
                                                 (int)round(1.0 * s->prm->smoothings), 1, 1);
      if (!v) return PLL_FAILURE;
      lh = -v;
    }
    if (s->st) s->st->insertions++;

    if (lh > entry->lh)
    {
      entry->lh = lh;
      entry->r = r;
      get_lens(s, p, entry->b1); get_lens(s, p->next, entry->b2); get_lens(s, p->next->next, entry->b3);
    }

    /* back to the lengths before the insertion, then take the subtree out again */
    set_lens(s, p, b1); set_lens(s, p->next, b2); set_lens(s, p->next->next, b3);
    invalidate_triplet(s, p);
    {
      pll_unode_t * gap = detach(s, p);
      set_lens(s, gap, regraft_len);
      pllhip_eval_invalidate_pmatrix(ev, gap);
    }

    descend = s->qdist[j] < s->prm->radius_max;
    if (s->cut && lh < s->cut->lh_start)
    {
      s->cut->lh_dec_count++;
      s->cut->lh_dec_sum += s->cut->lh_start - lh;
      descend = descend && (s->cut->lh_start - lh) < s->cut->lh_cutoff;
    }
    if (r->next && descend)
    {
      s->queue[count] = r->next->back;
      s->queue[count + 1] = r->next->next->back;
      s->qdist[count] = s->qdist[count + 1] = s->qdist[j] + 1;
      count += 2;
    }
  }

  TRACE("scan p=%u: %u candidates, best lh %.6f at r=%d\n", p->node_index, j, entry->lh,
        entry->r ? (int)entry->r->node_index : -1);
  /* back home, original lengths, everything looks at p again */
  attach(s, p, home);
  set_lens(s, p, z1); set_lens(s, p->next, z2); set_lens(s, p->next->next, z3);
  invalidate_triplet(s, p);
  root_at(s, p);
  pllhip_eval_invalidate_clv(ev, p);
  if (isnan(pllhip_eval_loglh(ev, 1))) return PLL_FAILURE;
  return PLL_SUCCESS;
}

/* algo_search.c:905-1050 */
static double scan_nodes(search_t * s, pll_unode_t ** nodes, unsigned int count)
{
  pllhip_eval_t * ev = s->ev;
  unsigned int i;
  double lh = pllhip_eval_loglh(ev, 0), best_lh = lh;
  undo_t * slot = s->ring + s->ring_cur;
  if (isnan(lh)) return 0.0;

  for (i = 0; i < count; ++i)
  {
    pll_unode_t * p = nodes[i], * r;
    placement_t entry;
    /* a two-taxon remainder has nowhere to go */
    if (!p->next->back->next && !p->next->next->back->next) continue;
    entry.p = p;
    entry.b1 = s->tmp + (size_t)9 * LV(s); entry.b2 = s->tmp + (size_t)10 * LV(s); entry.b3 = s->tmp + (size_t)11 * LV(s);
    if (s->cut) s->cut->lh_start = best_lh;
    if (!scan_placements(s, &entry)) return 0.0;
    r = entry.r;
    if (!r || r == p || r == p->back || r->back == p) continue;

    if (entry.lh - best_lh > 1e-6)
    {
      pll_unode_t * home = p->next->back;
      slot->p = p;
      slot->home = home;
      get_lens(s, p, slot->p_len);
      get_lens(s, p->next, slot->left_len);
      get_lens(s, p->next->next, slot->right_len);
      get_lens(s, r, slot->regraft_len);
      detach(s, p);
      attach(s, p, r);
      clamp_len(s, home);
      pllhip_eval_invalidate_pmatrix(ev, home);
      if (s->st)
      {
        if (s->st->log_count < PLLHIP_SPR_LOG_MAX)
        {
          s->st->log_prune[s->st->log_count] = p->node_index;
          s->st->log_regraft[s->st->log_count] = r->node_index;
          s->st->log_count++;
        }
        s->st->moves_applied++;
      }
      slot = ring_next(s);
      if (s->thorough)
      {
        set_lens(s, p, entry.b1); set_lens(s, p->next, entry.b2); set_lens(s, p->next->next, entry.b3);
      }
      else
      {
        clamp_len(s, p->next);
        clamp_len(s, p->next->next);
      }
      invalidate_triplet(s, p);
      pllhip_eval_invalidate_clv(ev, p);       /* the root is at p already */
      lh = pllhip_eval_loglh(ev, 1);
      if (isnan(lh)) return 0.0;
      TRACE("applied p=%u -> r=%u: lh %.6f (re-evaluated %.6f)\n", p->node_index, r->node_index,
            entry.lh, lh);
      best_lh = entry.lh;
    }
    else
    {
      entry.undo_pos = (unsigned int)ring_pos(s);
      best_save(s, &entry);
      lh = entry.lh;
    }
  }
  return lh;
}

double pllhip_eval_spr_round(pllhip_eval_t * ev, const pllhip_spr_params_t * prm,
                             pllhip_spr_cutoff_t * cut, pllhip_spr_stats_t * st)
{
  search_t s;
  undo_t redo;
  double * topol_lens = NULL;
  pll_unode_t ** nodes = NULL, * initial_root;
  saved_edge_t * best_topol = NULL;
  unsigned int i, k = 0, best_root = 0, nrec;
  double lh, best_lh, result = 0.0;
  size_t undone = 0;
  int li = -1;

  if (!ev || !prm || prm->radius_min < 1 || prm->radius_max < prm->radius_min || prm->bl_min <= 0 ||
      prm->bl_max < prm->bl_min || prm->smoothings < 1)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "Invalid SPR round parameters");
    return 0.0;
  }
  pll_errno = 0;
  memset(&s, 0, sizeof(s));
  memset(&redo, 0, sizeof(redo));
  s.ev = ev; s.prm = prm; s.cut = cut; s.st = st;
  s.thorough = prm->thorough;
  s.nl = (ev->linkage == PLLHIP_EVAL_BRLEN_UNLINKED && ev->part_brlens) ? 1 + ev->nparts : 1;
  s.ring_size = prm->ntopol_keep;
  s.best_size = prm->thorough ? prm->ntopol_keep : (size_t)prm->ntopol_keep * 3;
  nrec = ev->inner * 3;
  s.ring = (undo_t *)calloc(s.ring_size ? s.ring_size : 1, sizeof(undo_t));
  s.best = (placement_t *)calloc(s.best_size ? s.best_size : 1, sizeof(placement_t));
  {
    /* length vectors: 4 per remembered move (+ one spare move for re-applying a listed placement), 3 per listed
       placement, 1 per edge of the best topology, 12 scratch */
    const size_t ring_n = s.ring_size ? s.ring_size : 1, best_n = s.best_size ? s.best_size : 1;
    const size_t vectors = 4 * (ring_n + 1) + 3 * best_n + ev->edges + 12;
    s.pool = (double *)calloc(vectors * s.nl, sizeof(double));
    if (s.pool && s.ring && s.best)
    {
      double * v = s.pool;
      size_t x;
      for (x = 0; x < ring_n; ++x)
      {
        s.ring[x].p_len = v; v += s.nl; s.ring[x].left_len = v; v += s.nl;
        s.ring[x].right_len = v; v += s.nl; s.ring[x].regraft_len = v; v += s.nl;
      }
      redo.p_len = v; v += s.nl; redo.left_len = v; v += s.nl; redo.right_len = v; v += s.nl; redo.regraft_len = v; v += s.nl;
      for (x = 0; x < best_n; ++x) { s.best[x].b1 = v; v += s.nl; s.best[x].b2 = v; v += s.nl; s.best[x].b3 = v; v += s.nl; }
      topol_lens = v; v += (size_t)ev->edges * s.nl;
      s.tmp = v;
    }
  }
  s.records = (pll_unode_t **)calloc(ev->records, sizeof(*s.records));
  s.queue = (pll_unode_t **)calloc((size_t)ev->edges * 2 + 2, sizeof(*s.queue));
  s.qdist = (unsigned int *)calloc((size_t)ev->edges * 2 + 2, sizeof(unsigned int));
  nodes = (pll_unode_t **)calloc(nrec ? nrec : 1, sizeof(*nodes));
  best_topol = (saved_edge_t *)calloc(ev->edges, sizeof(saved_edge_t));
  if (!s.ring || !s.best || !s.records || !s.queue || !s.qdist || !nodes || !best_topol || !s.pool)
  {
    pllhip_eval_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate SPR round buffers");
    goto done;
  }
  for (i = 0; i < ev->edges; ++i) best_topol[i].length = topol_lens + (size_t)i * s.nl;
  for (i = 0; i < ev->tips + ev->inner; ++i)
  {
    pll_unode_t * n = ev->tree->nodes[i], * t = n;
    do { s.records[t->node_index] = t; t = t->next; } while (t && t != n);
  }
  if (st) memset(st, 0, sizeof(*st));
  if (cut) { cut->lh_dec_count = 0; cut->lh_dec_sum = 0.0; }

  lh = pllhip_eval_loglh(ev, 0);
  if (isnan(lh)) goto done;
  if (st) st->lnl_start = lh;
  initial_root = ev->root;

  list_records(ev->root->back, nodes, &k);
  list_records(ev->root, nodes, &k);
  lh = scan_nodes(&s, nodes, k);
  if (!lh) goto done;
  if (st) st->lnl_scan = lh;

  root_at(&s, initial_root);
  TRACE("scan done: lh %.6f\n", lh);
  best_lh = optimise_all(&s, prm->epsilon, 0.25);
  if (!best_lh) goto done;
  TRACE("after whole-tree optimisation: lh %.6f\n", best_lh);
  save_topology(&s, best_topol, &best_root);

  /* walk the history backwards: at every point in time first the placements that
     were scored but not applied then, then undo the move that led there */
  while (undone < s.ring_size)
  {
    const size_t pos = ring_pos(&s);
    int applied_listed = 0;
    li = best_next(&s, pos, li);
    if (li < 0)
    {
      undo_t * u = ring_prev(&s);
      pll_unode_t * cur1, * cur2;
      if (!u || !u->p) break;
      cur1 = u->p->next->back;
      cur2 = u->home->back;
      detach(&s, u->p);
      attach(&s, u->p, u->home);
      set_lens(&s, cur1, u->regraft_len);
      set_lens(&s, u->p, u->p_len);
      set_lens(&s, u->home, u->left_len);
      set_lens(&s, cur2, u->right_len);
      undone++;
    }
    else
    {
      placement_t * e;
      if ((unsigned int)li > prm->ntopol_keep) continue;
      e = &s.best[li];
      redo.p = e->p;
      redo.home = e->p->next->back;
      get_lens(&s, e->p, redo.p_len);
      get_lens(&s, e->p->next, redo.left_len);
      get_lens(&s, e->p->next->next, redo.right_len);
      get_lens(&s, e->r, redo.regraft_len);
      detach(&s, e->p);
      attach(&s, e->p, e->r);
      clamp_len(&s, redo.home);
      if (prm->thorough)
      {
        set_lens(&s, e->p, e->b1); set_lens(&s, e->p->next, e->b2); set_lens(&s, e->p->next->next, e->b3);
      }
      else
      {
        clamp_len(&s, e->p); clamp_len(&s, e->p->next); clamp_len(&s, e->p->next->next);
      }
      applied_listed = 1;
    }

    lh = optimise_all(&s, prm->epsilon, 0.25);
    if (!lh) goto done;
    if (st) st->rescored++;
    TRACE("rescored (%s, history %zu): lh %.6f, best %.6f\n", applied_listed ? "listed" : "undo", pos, lh, best_lh);
    if (lh - best_lh > 0.01)
    {
      save_topology(&s, best_topol, NULL);
      best_lh = lh;
    }
    if (applied_listed)
    {
      pll_unode_t * cur1 = redo.p->next->back, * cur2 = redo.home->back;
      detach(&s, redo.p);
      attach(&s, redo.p, redo.home);
      set_lens(&s, cur1, redo.regraft_len);
      set_lens(&s, redo.p, redo.p_len);
      set_lens(&s, redo.home, redo.left_len);
      set_lens(&s, cur2, redo.right_len);
    }
  }

  if (cut) cut->lh_cutoff = prm->subtree_cutoff * (cut->lh_dec_sum / cut->lh_dec_count);

  load_topology(&s, best_topol, best_root);
  lh = pllhip_eval_loglh(ev, 0);
  if (isnan(lh)) goto done;
  if (fabs(lh - best_lh) > 1e-6)
  {
    pllhip_eval_error(PLL_ERROR_PARAM_INVALID, "SPR round: restored tree scores %.12f, expected %.12f",
                      lh, best_lh);
    goto done;
  }
  if (st) st->lnl_final = lh;
  result = lh;

done:
  free(s.ring); free(s.best); free(s.records); free(s.queue); free(s.qdist); free(s.pool);
  free(nodes); free(best_topol);
  return result;
}
