/*
 * pll_utree_moves.c -- SPR and NNI on the pll_unode_t graph, with rollback.
 * Host pointer surgery only.  Contract as pll-modules uses it:
 *
 *   pll_utree_spr(p, r, rb, branch_lengths, matrix_indices)
 *     (src/tree/pll_tree.c:186): p is a record of an inner node; the subtree
 *     behind p is pruned (its two other neighbours u = p->next->back and
 *     v = p->next->next->back are joined, lengths added) and re-inserted into
 *     the edge r <-> r->back, which is split in half.  Afterwards
 *     p->next->back == r and p->next->next->back == old r->back -- the state
 *     utree_rollback_spr relies on (src/tree/pll_tree.c:1894-1914: undoing is
 *     another SPR of p onto old u, then the four lengths are restored).
 *     P-matrix indices stay a permutation of the edge set: (u,v) keeps u's,
 *     (r', p->next->next) keeps p->next->next's, (r, p->next) takes r's.
 *   pll_utree_nni(p, type, rb) (src/tree/pll_tree.c:241): swaps the subtree at
 *     p->next with the one at p->back->next (LEFT) or p->back->next->next
 *     (RIGHT); each subtree keeps the length and P-matrix index of the edge it
 *     moves to.  Applying the same move again restores the tree
 *     (src/tree/pll_tree.c:1917-1937).
 */
#include "pll.h"
#include <stdarg.h>

static void moves_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, 200, fmt, ap);
  va_end(ap);
}

static void join(pll_unode_t * a, pll_unode_t * b, double length, unsigned int pmatrix_index)
{
  a->back = b;
  b->back = a;
  a->length = b->length = length;
  a->pmatrix_index = b->pmatrix_index = pmatrix_index;
}

int pll_utree_spr(pll_unode_t * p, pll_unode_t * r, pll_utree_rb_t * rb,
                  double * branch_lengths, unsigned int * matrix_indices)
{
  if ((branch_lengths == NULL) != (matrix_indices == NULL))
  {
    moves_error(PLL_ERROR_PARAM_INVALID, "Parameters 4,5 must be both NULL or both set");
    return PLL_FAILURE;
  }
  if (!p->next)
  {
    moves_error(PLL_ERROR_SPR_TERMINALBRANCH, "Prune edge must be defined by an inner node");
    return PLL_FAILURE;
  }
  pll_unode_t * q = p->next, * q2 = p->next->next;
  if (r == p || r == p->back || r == q || r == q->back || r == q2 || r == q2->back)
  {
    moves_error(PLL_ERROR_SPR_NOCHANGE, "Proposed move yields the same tree");
    return PLL_FAILURE;
  }
  pll_unode_t * u = q->back, * v = q2->back, * r2 = r->back;
  if (rb)
  {
    rb->move_type = PLL_UTREE_MOVE_SPR;
    rb->SPR.p = p;
    rb->SPR.r = r;
    rb->SPR.rb = r2;
    rb->SPR.r_len = r->length;
    rb->SPR.pnb = u;
    rb->SPR.pnb_len = q->length;
    rb->SPR.pnnb = v;
    rb->SPR.pnnb_len = q2->length;
  }
  int k = 0;
  /* close the gap left by the pruned subtree */
  join(u, v, u->length + v->length, u->pmatrix_index);
  if (branch_lengths) { branch_lengths[k] = u->length; matrix_indices[k++] = u->pmatrix_index; }
  /* open the regraft edge and insert */
  const double half = r->length / 2;
  join(r2, q2, half, q2->pmatrix_index);
  if (branch_lengths) { branch_lengths[k] = half; matrix_indices[k++] = q2->pmatrix_index; }
  join(r, q, half, r->pmatrix_index);
  if (branch_lengths) { branch_lengths[k] = half; matrix_indices[k++] = r->pmatrix_index; }
  return PLL_SUCCESS;
}

static int in_subtree(const pll_unode_t * root, const pll_unode_t * node)
{
  if (root == node) return 1;
  if (!root->next) return 0;
  for (const pll_unode_t * s = root->next; s != root; s = s->next)
  {
    if (s == node) return 1;
    if (in_subtree(s->back, node)) return 1;
  }
  return 0;
}

int pll_utree_spr_safe(pll_unode_t * p, pll_unode_t * r, pll_utree_rb_t * rb,
                       double * branch_lengths, unsigned int * matrix_indices)
{
  if (!p->next)
  {
    moves_error(PLL_ERROR_SPR_TERMINALBRANCH, "Prune edge must be defined by an inner node");
    return PLL_FAILURE;
  }
  /* the regraft edge must not lie inside the pruned subtree (behind p) */
  if (in_subtree(p->back, r) || in_subtree(p->back, r->back))
  {
    moves_error(PLL_ERROR_SPR_NOCHANGE, "Regraft edge is part of the pruned subtree");
    return PLL_FAILURE;
  }
  return pll_utree_spr(p, r, rb, branch_lengths, matrix_indices);
}

int pll_utree_nni(pll_unode_t * p, int type, pll_utree_rb_t * rb)
{
  if (type != PLL_UTREE_MOVE_NNI_LEFT && type != PLL_UTREE_MOVE_NNI_RIGHT)
  {
    moves_error(PLL_ERROR_NNI_INVALIDMOVE, "Invalid NNI move type");
    return PLL_FAILURE;
  }
  if (!p->next || !p->back->next)
  {
    moves_error(PLL_ERROR_NNI_TERMINALBRANCH, "Specified terminal branch");
    return PLL_FAILURE;
  }
  if (rb)
  {
    rb->move_type = PLL_UTREE_MOVE_NNI;
    rb->NNI.p = p;
    rb->NNI.nni_type = type;
  }
  pll_unode_t * s1 = p->next;
  pll_unode_t * s2 = (type == PLL_UTREE_MOVE_NNI_LEFT) ? p->back->next : p->back->next->next;
  /* exchange the subtrees hanging off s1 and s2; the edges keep their
     lengths and P-matrix indices */
  pll_unode_t * t1 = s1->back, * t2 = s2->back;
  const double l1 = s1->length, l2 = s2->length;
  const unsigned int m1 = s1->pmatrix_index, m2 = s2->pmatrix_index;
  join(s1, t2, l2, m2);
  join(s2, t1, l1, m1);
  return PLL_SUCCESS;
}

int pll_utree_rollback(pll_utree_rb_t * rb, double * branch_lengths,
                       unsigned int * matrix_indices)
{
  if ((branch_lengths == NULL) != (matrix_indices == NULL))
  {
    moves_error(PLL_ERROR_PARAM_INVALID, "Parameters 2,3 must be both NULL or both set");
    return PLL_FAILURE;
  }
  if (rb->move_type == PLL_UTREE_MOVE_SPR)
  {
    pll_unode_t * p = rb->SPR.p, * q = p->next, * q2 = p->next->next;
    pll_unode_t * r = rb->SPR.r, * u = rb->SPR.pnb, * v = rb->SPR.pnnb;
    int k = 0;
    join(r, rb->SPR.rb, rb->SPR.r_len, r->pmatrix_index);
    if (branch_lengths) { branch_lengths[k] = rb->SPR.r_len; matrix_indices[k++] = r->pmatrix_index; }
    join(q, u, rb->SPR.pnb_len, u->pmatrix_index);
    if (branch_lengths) { branch_lengths[k] = rb->SPR.pnb_len; matrix_indices[k++] = u->pmatrix_index; }
    join(q2, v, rb->SPR.pnnb_len, q2->pmatrix_index);
    if (branch_lengths) { branch_lengths[k] = rb->SPR.pnnb_len; matrix_indices[k++] = q2->pmatrix_index; }
    return PLL_SUCCESS;
  }
  if (rb->move_type == PLL_UTREE_MOVE_NNI)
    return pll_utree_nni(rb->NNI.p, rb->NNI.nni_type, NULL);
  moves_error(PLL_ERROR_PARAM_INVALID, "Invalid rollback record");
  return PLL_FAILURE;
}
