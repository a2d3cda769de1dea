/*
 * pll_utree.c -- unrooted-tree utilities of the boundary (tier B2 of
 * include/pll.h): post-order traversal with a skip callback, conversion of a
 * traversal into a pll_operation_t list, wrap / clone / destroy, newick
 * import/export.  Host pointer code only; no likelihood arithmetic.
 *
 * Contracts reconstructed from the reference call sites:
 *   traverse(root, POSTORDER, cb, buf, &n): root must be an inner record; a
 *     full traversal of an n-tip binary tree yields 2n-2 records, the root
 *     last (src/tree/treeinfo.c:973-992); cb()==0 skips the node and its
 *     subtree (src/tree/treeinfo.c:38-61).
 *   create_operations(buf, n, branches|NULL, pmatrix_indices|NULL, ops,
 *     matrix_count|NULL, &ops_count) (src/tree/treeinfo.c:1009-1015): one op
 *     per inner record, children = next->back and next->next->back, the op
 *     field order of src/optimize/pll_optimize.c:758-765.
 *   wraptree(root, tips): pll_utree_t with tips first, then one record per
 *     inner ring (src/tree/treeinfo.c:1241-1260, 1320).
 */
#include "pll.h"
#include <ctype.h>
#include <stdarg.h>

static void utree_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, 200, fmt, ap);
  va_end(ap);
}

/* ------------------------------------------------------------------ */
/* traversal                                                          */
/* ------------------------------------------------------------------ */

static void walk(pll_unode_t * node, int order, int (*cb)(pll_unode_t *),
                 pll_unode_t ** out, unsigned int * count)
{
  if (!cb(node)) return;
  if (order == PLL_TREE_TRAVERSE_PREORDER) out[(*count)++] = node;
  if (node->next)
  {
    pll_unode_t * s = node->next;
    while (s != node)
    {
      walk(s->back, order, cb, out, count);
      s = s->next;
    }
  }
  if (order == PLL_TREE_TRAVERSE_POSTORDER) out[(*count)++] = node;
}

int pll_utree_traverse(pll_unode_t * root, int traversal,
                       int (*cbtrav)(pll_unode_t *),
                       pll_unode_t ** outbuffer, unsigned int * trav_size)
{
  *trav_size = 0;
  if (!root->next)
  {
    utree_error(PLL_ERROR_PARAM_INVALID, "Traversal root must be an inner node");
    return PLL_FAILURE;
  }
  if (traversal != PLL_TREE_TRAVERSE_POSTORDER &&
      traversal != PLL_TREE_TRAVERSE_PREORDER)
  {
    utree_error(PLL_ERROR_PARAM_INVALID, "Invalid traversal value");
    return PLL_FAILURE;
  }
  if (traversal == PLL_TREE_TRAVERSE_POSTORDER)
  {
    walk(root->back, traversal, cbtrav, outbuffer, trav_size);
    walk(root, traversal, cbtrav, outbuffer, trav_size);
  }
  else
  {
    walk(root, traversal, cbtrav, outbuffer, trav_size);
    walk(root->back, traversal, cbtrav, outbuffer, trav_size);
  }
  return PLL_SUCCESS;
}

void pll_utree_create_operations(pll_unode_t * const * trav, unsigned int n,
                                 double * branches,
                                 unsigned int * pmatrix_indices,
                                 pll_operation_t * ops,
                                 unsigned int * matrix_count,
                                 unsigned int * ops_count)
{
  unsigned int i, nm = 0, no = 0;
  const pll_unode_t * last = n ? trav[n - 1] : NULL;
  for (i = 0; i < n; ++i)
  {
    const pll_unode_t * node = trav[i];
    /* the edge between the virtual root and root->back is listed once */
    if (!(last && node == last->back))
    {
      if (branches) branches[nm] = node->length;
      if (pmatrix_indices) pmatrix_indices[nm] = node->pmatrix_index;
      nm++;
    }
    if (node->next)
    {
      const pll_unode_t * c1 = node->next->back;
      const pll_unode_t * c2 = node->next->next->back;
      pll_operation_t * op = &ops[no++];
      op->parent_clv_index = node->clv_index;
      op->parent_scaler_index = node->scaler_index;
      op->child1_clv_index = c1->clv_index;
      op->child1_scaler_index = c1->scaler_index;
      op->child1_matrix_index = c1->pmatrix_index;
      op->child2_clv_index = c2->clv_index;
      op->child2_scaler_index = c2->scaler_index;
      op->child2_matrix_index = c2->pmatrix_index;
    }
  }
  if (matrix_count) *matrix_count = nm;
  if (ops_count) *ops_count = no;
}

/* ------------------------------------------------------------------ */
/* wrap / destroy / clone                                             */
/* ------------------------------------------------------------------ */

static int cb_all(pll_unode_t * node) { (void)node; return 1; }

static unsigned int count_records(pll_unode_t * node, unsigned int * tips,
                                  unsigned int * inner)
{
  /* counts nodes in the subtree behind `node` (node included) */
  unsigned int c = 1;
  if (!node->next) { (*tips)++; return 1; }
  (*inner)++;
  for (pll_unode_t * s = node->next; s != node; s = s->next)
    c += count_records(s->back, tips, inner);
  return c;
}

pll_utree_t * pll_utree_wraptree_multi(pll_unode_t * root,
                                       unsigned int tip_count,
                                       unsigned int inner_count)
{
  unsigned int i, n, tips = 0, inner = 0;
  if (!root->next) root = root->back;
  if (!root || !root->next)
  {
    utree_error(PLL_ERROR_PARAM_INVALID, "The tree has no inner nodes");
    return NULL;
  }
  count_records(root->back, &tips, &inner);
  count_records(root, &tips, &inner);
  if (tip_count == 0) tip_count = tips;
  if (tip_count != tips || (inner_count && inner_count != inner))
  {
    utree_error(PLL_ERROR_PARAM_INVALID,
                "tip/inner count (%u/%u) does not match the tree (%u/%u)",
                tip_count, inner_count, tips, inner);
    return NULL;
  }
  pll_utree_t * tree = (pll_utree_t *)calloc(1, sizeof(*tree));
  pll_unode_t ** buf = (pll_unode_t **)calloc(tips + inner, sizeof(*buf));
  if (tree) tree->nodes = (pll_unode_t **)calloc(tips + inner, sizeof(*buf));
  if (!tree || !buf || !tree->nodes)
  {
    free(buf); if (tree) free(tree->nodes); free(tree);
    utree_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return NULL;
  }
  pll_utree_traverse(root, PLL_TREE_TRAVERSE_POSTORDER, cb_all, buf, &n);

  /* tips go to the slot named by their node_index when the indices form a
     permutation of 0..tips-1 (the usual case), otherwise in visiting order */
  int permutation = 1;
  for (i = 0; i < n && permutation; ++i)
    if (!buf[i]->next)
    {
      unsigned int k = buf[i]->node_index;
      if (k >= tips || tree->nodes[k]) permutation = 0;
      else tree->nodes[k] = buf[i];
    }
  if (!permutation) memset(tree->nodes, 0, sizeof(*buf) * tips);
  unsigned int t = 0, in = tips;
  for (i = 0; i < n; ++i)
  {
    if (buf[i]->next) tree->nodes[in++] = buf[i];
    else if (!permutation) tree->nodes[t++] = buf[i];
  }
  free(buf);

  tree->tip_count = tips;
  tree->inner_count = inner;
  tree->edge_count = tips + inner - 1;
  tree->binary = (inner == tips - 2);
  tree->vroot = root;
  return tree;
}

pll_utree_t * pll_utree_wraptree(pll_unode_t * root, unsigned int tip_count)
{
  return pll_utree_wraptree_multi(root, tip_count, 0);
}

static void free_ring(pll_unode_t * node, void (*cb_destroy)(void *))
{
  if (!node) return;
  pll_unode_t * s = node->next;
  if (cb_destroy && node->data) cb_destroy(node->data);
  free(node->label);
  while (s && s != node)
  {
    pll_unode_t * nx = s->next;
    free(s);
    s = nx;
  }
  free(node);
}

void pll_utree_destroy(pll_utree_t * tree, void (*cb_destroy)(void *))
{
  unsigned int i;
  if (!tree) return;
  for (i = 0; i < tree->tip_count + tree->inner_count; ++i)
    free_ring(tree->nodes[i], cb_destroy);
  free(tree->nodes);
  free(tree);
}

static void graph_destroy_rec(pll_unode_t * node, void (*cb_destroy)(void *))
{
  if (node->next)
    for (pll_unode_t * s = node->next; s != node; s = s->next)
      graph_destroy_rec(s->back, cb_destroy);
  free_ring(node, cb_destroy);
}

void pll_utree_graph_destroy(pll_unode_t * root, void (*cb_destroy)(void *))
{
  if (!root) return;
  if (!root->next) root = root->back;
  pll_unode_t * other = root->back;
  graph_destroy_rec(root, cb_destroy);
  if (other) graph_destroy_rec(other, cb_destroy);
}

static pll_unode_t * copy_record(const pll_unode_t * src)
{
  pll_unode_t * n = (pll_unode_t *)calloc(1, sizeof(*n));
  if (!n) return NULL;
  *n = *src;
  n->next = n->back = NULL;
  n->label = NULL;
  return n;
}

/* clone the subtree hanging behind `src` (src itself included); the returned
   record corresponds to src and has back == NULL */
static pll_unode_t * clone_rec(const pll_unode_t * src)
{
  pll_unode_t * head = copy_record(src);
  if (!head) return NULL;
  if (src->label) head->label = strdup(src->label);
  if (src->next)
  {
    pll_unode_t * prev = head;
    for (const pll_unode_t * s = src->next; s != src; s = s->next)
    {
      pll_unode_t * r = copy_record(s);
      if (!r) return NULL;
      r->label = head->label;
      prev->next = r;
      prev = r;
      pll_unode_t * child = clone_rec(s->back);
      if (!child) return NULL;
      r->back = child;
      child->back = r;
    }
    prev->next = head;
  }
  return head;
}

pll_unode_t * pll_utree_graph_clone(const pll_unode_t * root)
{
  if (!root->next) root = root->back;
  pll_unode_t * a = clone_rec(root);
  pll_unode_t * b = a ? clone_rec(root->back) : NULL;
  if (!a || !b)
  {
    utree_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return NULL;
  }
  a->back = b;
  b->back = a;
  return a;
}

pll_utree_t * pll_utree_clone(const pll_utree_t * tree)
{
  pll_unode_t * root = pll_utree_graph_clone(tree->vroot);
  if (!root) return NULL;
  return pll_utree_wraptree_multi(root, tree->tip_count, tree->inner_count);
}

int pll_utree_every(pll_utree_t * tree, int (*cb)(pll_unode_t *))
{
  unsigned int i;
  int rc = 1;
  for (i = 0; i < tree->tip_count + tree->inner_count; ++i)
  {
    pll_unode_t * n = tree->nodes[i], * s = n;
    do { rc &= cb(s); s = s->next; } while (s && s != n);
  }
  return rc ? PLL_SUCCESS : PLL_FAILURE;
}

static int check_record(pll_unode_t * n)
{
  if (!n->back || n->back->back != n) return 0;
  if (n->pmatrix_index != n->back->pmatrix_index) return 0;
  if (fabs(n->length - n->back->length) > 1e-12) return 0;
  if (n->next)
  {
    if (n->next->clv_index != n->clv_index) return 0;
    if (n->next->scaler_index != n->scaler_index) return 0;
  }
  return 1;
}

int pll_utree_check_integrity(const pll_utree_t * tree)
{
  if (!pll_utree_every((pll_utree_t *)tree, check_record))
  {
    utree_error(PLL_ERROR_TREE_INVALID, "Inconsistent back pointers, lengths or indices");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

/* canonical indices: tip t -> clv t, pmatrix t, no scaler; inner ring k (post
   order) -> clv tips+k, scaler k, node_index tips+3k+{0,1,2}; inner-inner
   edges get pmatrix indices from `tips` upwards */
void pll_utree_reset_template_indices(pll_unode_t * root, unsigned int tip_count)
{
  unsigned int n, i, inner = 0, edge = tip_count;
  if (!root->next) root = root->back;
  pll_unode_t ** buf = (pll_unode_t **)calloc(2 * tip_count, sizeof(*buf));
  if (!buf) return;
  pll_utree_traverse(root, PLL_TREE_TRAVERSE_POSTORDER, cb_all, buf, &n);
  for (i = 0; i < n; ++i)
  {
    pll_unode_t * node = buf[i];
    if (!node->next)
    {
      node->clv_index = node->node_index;
      node->scaler_index = PLL_SCALE_BUFFER_NONE;
      node->pmatrix_index = node->back->pmatrix_index = node->node_index;
    }
    else
    {
      unsigned int k = 0;
      pll_unode_t * s = node;
      do
      {
        s->clv_index = tip_count + inner;
        s->scaler_index = (int)inner;
        s->node_index = tip_count + 3 * inner + k++;
        s = s->next;
      } while (s != node);
      inner++;
    }
  }
  for (i = 0; i < n; ++i)
  {
    pll_unode_t * node = buf[i];
    if (node->next && node->back->next && node != root->back && node != root)
      node->pmatrix_index = node->back->pmatrix_index = edge++;
  }
  if (root->back->next)
    root->pmatrix_index = root->back->pmatrix_index = edge++;
  free(buf);
}

/* ------------------------------------------------------------------ */
/* newick                                                             */
/* ------------------------------------------------------------------ */

typedef struct { char * s; size_t len, cap; } sbuf_t;

static void sb_put(sbuf_t * b, const char * t)
{
  size_t n = strlen(t);
  if (b->len + n + 1 > b->cap)
  {
    b->cap = (b->len + n + 1) * 2;
    b->s = (char *)realloc(b->s, b->cap);
  }
  memcpy(b->s + b->len, t, n + 1);
  b->len += n;
}

static void export_rec(const pll_unode_t * node, sbuf_t * b,
                       char * (*cb)(const pll_unode_t *))
{
  char tmp[64];
  if (node->next)
  {
    sb_put(b, "(");
    for (const pll_unode_t * s = node->next; s != node; s = s->next)
    {
      export_rec(s->back, b, cb);
      if (s->next != node) sb_put(b, ",");
    }
    sb_put(b, ")");
  }
  if (cb)
  {
    char * t = cb(node);
    sb_put(b, t);
    free(t);
  }
  else
  {
    if (node->label) sb_put(b, node->label);
    snprintf(tmp, sizeof(tmp), ":%f", node->length);
    sb_put(b, tmp);
  }
}

/* "(subtree of root->back, children of root...);" -- the form printed in
   test/out/optimize/blopt-minimal.out:68 */
char * pll_utree_export_newick(const pll_unode_t * root,
                               char * (*cb_serialize)(const pll_unode_t *))
{
  sbuf_t b = {NULL, 0, 0};
  if (!root) return NULL;
  if (!root->next) root = root->back;
  sb_put(&b, "(");
  export_rec(root->back, &b, cb_serialize);
  for (const pll_unode_t * s = root->next; s != root; s = s->next)
  {
    sb_put(&b, ",");
    export_rec(s->back, &b, cb_serialize);
  }
  sb_put(&b, ")");
  if (root->label) sb_put(&b, root->label);
  sb_put(&b, ";");
  return b.s;
}

typedef struct { const char * p; unsigned int tips; int err; } nwk_t;

static void nwk_skip(nwk_t * k) { while (isspace((unsigned char)*k->p)) k->p++; }

static char * nwk_label(nwk_t * k)
{
  nwk_skip(k);
  const char * a = k->p;
  if (*a == '\'')
  {
    a = ++k->p;
    while (*k->p && *k->p != '\'') k->p++;
    char * l = strndup(a, (size_t)(k->p - a));
    if (*k->p) k->p++;
    return l;
  }
  while (*k->p && !strchr("(),:;[", *k->p) && !isspace((unsigned char)*k->p)) k->p++;
  return (k->p > a) ? strndup(a, (size_t)(k->p - a)) : NULL;
}

static double nwk_length(nwk_t * k)
{
  nwk_skip(k);
  if (*k->p != ':') return 0.0;
  k->p++;
  char * end;
  double v = strtod(k->p, &end);
  if (end == k->p) k->err = 1;
  k->p = end;
  return v;
}

/* parses one subtree; returns the record that faces the parent (back unset) */
static pll_unode_t * nwk_subtree(nwk_t * k)
{
  nwk_skip(k);
  pll_unode_t * head = (pll_unode_t *)calloc(1, sizeof(*head));
  if (!head) { k->err = 1; return NULL; }
  head->scaler_index = PLL_SCALE_BUFFER_NONE;
  if (*k->p == '(')
  {
    pll_unode_t * prev = head;
    k->p++;
    for (;;)
    {
      pll_unode_t * child = nwk_subtree(k);
      if (!child) { k->err = 1; return head; }
      pll_unode_t * r = (pll_unode_t *)calloc(1, sizeof(*r));
      if (!r) { k->err = 1; return head; }
      r->back = child;
      child->back = r;
      r->length = child->length;
      prev->next = r;
      prev = r;
      nwk_skip(k);
      if (*k->p == ',') { k->p++; continue; }
      if (*k->p == ')') { k->p++; break; }
      k->err = 1;
      return head;
    }
    prev->next = head;
    head->label = nwk_label(k);
    for (pll_unode_t * s = head->next; s != head; s = s->next) s->label = head->label;
  }
  else
  {
    head->label = nwk_label(k);
    if (!head->label) k->err = 1;
    head->node_index = head->clv_index = k->tips++;
  }
  head->length = nwk_length(k);
  return head;
}

static pll_utree_t * parse_string(const char * s, int unroot)
{
  nwk_t k = {s, 0, 0};
  pll_unode_t * top = nwk_subtree(&k);
  nwk_skip(&k);
  if (k.err || !top || !top->next || *k.p != ';')
  {
    utree_error(PLL_ERROR_NEWICK_SYNTAX, "Newick syntax error near offset %ld",
                (long)(k.p - s));
    return NULL;
  }
  /* `top` is an extra record of the root ring facing a non-existent parent:
     unlink it */
  unsigned int degree = 0;
  pll_unode_t * last = top;
  for (pll_unode_t * r = top->next; r != top; r = r->next) { degree++; last = r; }
  pll_unode_t * root = top->next;
  last->next = root;
  char * root_label = top->label;
  free(top);
  if (degree == 2)
  {
    if (!unroot)
    {
      utree_error(PLL_ERROR_TREE_CONVERSION, "The tree is rooted (binary root)");
      return NULL;
    }
    /* join the two root children into one edge */
    pll_unode_t * a = root->back, * b = root->next->back;
    double len = a->length + b->length;
    a->back = b; b->back = a;
    a->length = b->length = len;
    free(root->next);
    free(root);
    free(root_label);
    root = a->next ? a : b;
    if (!root->next)
    {
      utree_error(PLL_ERROR_TREE_CONVERSION, "A tree needs at least three tips");
      return NULL;
    }
  }
  else
    for (pll_unode_t * r = root;;)
    {
      r->label = root_label;
      r = r->next;
      if (r == root) break;
    }
  pll_utree_reset_template_indices(root, k.tips);
  return pll_utree_wraptree(root, k.tips);
}

pll_utree_t * pll_utree_parse_newick_string(const char * s)
{
  return parse_string(s, 0);
}

pll_utree_t * pll_utree_parse_newick_string_unroot(const char * s)
{
  return parse_string(s, 1);
}

static pll_utree_t * parse_file(const char * filename, int unroot)
{
  FILE * f = fopen(filename, "rb");
  if (!f)
  {
    utree_error(PLL_ERROR_FILE_OPEN, "Unable to open file (%s)", filename);
    return NULL;
  }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char * buf = (char *)malloc((size_t)sz + 1);
  if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz)
  {
    fclose(f); free(buf);
    utree_error(PLL_ERROR_FILE_EOF, "Unable to read file (%s)", filename);
    return NULL;
  }
  buf[sz] = 0;
  fclose(f);
  pll_utree_t * t = parse_string(buf, unroot);
  free(buf);
  return t;
}

pll_utree_t * pll_utree_parse_newick(const char * filename)
{
  return parse_file(filename, 0);
}

pll_utree_t * pll_utree_parse_newick_unroot(const char * filename)
{
  return parse_file(filename, 1);
}

static void ascii_rec(const pll_unode_t * node, int depth, int options)
{
  printf("%*s+-- ", depth * 4, "");
  if ((options & PLL_UTREE_SHOW_LABEL) && node->label) printf("%s", node->label);
  if (options & PLL_UTREE_SHOW_BRANCH_LENGTH) printf(" :%f", node->length);
  if (options & PLL_UTREE_SHOW_CLV_INDEX) printf(" clv:%u", node->clv_index);
  if (options & PLL_UTREE_SHOW_SCALER_INDEX) printf(" sc:%d", node->scaler_index);
  if (options & PLL_UTREE_SHOW_PMATRIX_INDEX) printf(" pm:%u", node->pmatrix_index);
  printf("\n");
  if (node->next)
    for (const pll_unode_t * s = node->next; s != node; s = s->next)
      ascii_rec(s->back, depth + 1, options);
}

void pll_utree_show_ascii(const pll_unode_t * root, int options)
{
  if (!root->next) root = root->back;
  ascii_rec(root->back, 0, options);
  for (const pll_unode_t * s = root->next; s != root; s = s->next)
    ascii_rec(s->back, 0, options);
}
