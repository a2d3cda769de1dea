/*
 * treeinfo_driver.c -- a small client of pll-modules' upper layers, written for
 * this repository's drop-in test (tests/test_dropin_modules.py).  It only calls
 * PUBLIC pll-modules functions (pllmod_treeinfo_*, pllmod_algo_*) and libpll
 * functions of include/pll.h; the pll-modules sources themselves are compiled
 * unchanged from /root/reference by the test and linked against this
 * repository's library.
 *
 * Scenario: 12 taxa, two partitions (DNA GTR+G4, 5-state GTR+G4), linked branch
 * lengths.  Prints: full lnL, incremental lnL after invalidating one CLV, lnL
 * after branch-length optimisation, lnL after one SPR round, and the final full
 * recomputation.
 */
#include "pllmod_common.h"
#include "pll_tree.h"
#include "pll_optimize.h"
#include "pllmod_algorithm.h"
#include "pllhip_eval.h"
#include <stdio.h>

#define TAXA 12

/* pll-modules' newick-split parser is generated from src/tree/lex_split.l +
   split_utree.y by flex/bison, which this image lacks; consensus.c (pulled in
   through the constraint checker) references it.  Nothing in this scenario
   reaches it. */
pll_split_t * pll_utree_split_newick_string(char * s, unsigned int tip_count,
                                            string_hashtable_t * names_hash)
{
  (void)s; (void)tip_count; (void)names_hash;
  fprintf(stderr, "pll_utree_split_newick_string: parser not built in this test\n");
  abort();
}

static unsigned long long rng_state = 0x1234567ULL;
static unsigned int rnd(unsigned int n)
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (unsigned int)((rng_state >> 33) % n);
}

static pll_partition_t * make_partition(const pll_utree_t * tree, unsigned int states,
                                        unsigned int sites, unsigned int attrs)
{
  unsigned int i, s;
  pll_partition_t * p = pll_partition_create(TAXA, TAXA - 2, states, sites, 1, 2 * TAXA - 3, 4,
                                             TAXA - 2, attrs);
  if (!p) { fprintf(stderr, "partition: %s\n", pll_errmsg); exit(1); }
  double freqs[5], rates[10];
  for (i = 0; i < states; ++i) freqs[i] = 1.0 / states;
  for (i = 0; i < states * (states - 1) / 2; ++i) rates[i] = 0.5 + 0.25 * (i % 5);
  rates[states * (states - 1) / 2 - 1] = 1.0;
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, rates);
  double cats[4];
  pll_compute_gamma_cats(0.7, 4, cats, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, cats);

  pll_state_t map[256];
  memset(map, 0, sizeof(map));
  for (s = 0; s < states; ++s) map['a' + s] = 1ULL << s;
  map['-'] = (1ULL << states) - 1;
  /* a weakly structured alignment: a random column, mutated per taxon */
  char * seq[TAXA];
  for (i = 0; i < TAXA; ++i) seq[i] = (char *)malloc(sites + 1);
  for (s = 0; s < sites; ++s)
  {
    char base = (char)('a' + rnd(states));
    for (i = 0; i < TAXA; ++i)
      seq[i][s] = (rnd(10) < 3) ? (char)('a' + rnd(states)) : ((rnd(40) == 0) ? '-' : base);
  }
  for (i = 0; i < TAXA; ++i)
  {
    seq[i][sites] = 0;
    unsigned int tip = tree->nodes[i]->clv_index;
    if (!pll_set_tip_states(p, tip, map, seq[i])) { fprintf(stderr, "tips: %s\n", pll_errmsg); exit(1); }
    free(seq[i]);
  }
  return p;
}

static int cb_all(pll_unode_t * n) { (void)n; return 1; }

/* the same likelihood through raw include/pll.h calls only (no treeinfo) */
static double direct_lnl(pll_utree_t * tree, pll_partition_t ** parts, unsigned int nparts)
{
  unsigned int n = 0, nm = 0, no = 0, k, q;
  pll_unode_t * trav[2 * TAXA];
  double brlens[2 * TAXA];
  unsigned int midx[2 * TAXA], params[4] = {0, 0, 0, 0};
  pll_operation_t ops[TAXA];
  pll_unode_t * root = tree->vroot;
  double total = 0.0;
  if (!pll_utree_traverse(root, PLL_TREE_TRAVERSE_POSTORDER, cb_all, trav, &n)) exit(2);
  pll_utree_create_operations(trav, n, brlens, midx, ops, &nm, &no);
  for (q = 0; q < nparts; ++q)
  {
    for (k = 0; k < nm; ++k)          /* one call per branch, like treeinfo */
      if (!pll_update_prob_matrices(parts[q], params, &midx[k], &brlens[k], 1)) exit(3);
    pll_update_partials(parts[q], ops, no);
    total += pll_compute_edge_loglikelihood(parts[q], root->clv_index, root->scaler_index,
                                            root->back->clv_index, root->back->scaler_index,
                                            root->pmatrix_index, params, NULL);
  }
  return total;
}

int main(int argc, char ** argv)
{
  unsigned int attrs = (argc > 1 && !strcmp(argv[1], "tv")) ? PLL_ATTRIB_PATTERN_TIP : 0;
  const char * nwk = "((t0:0.11,t1:0.07):0.05,(t2:0.13,(t3:0.06,t4:0.09):0.04):0.03,"
                     "((t5:0.10,(t6:0.05,t7:0.12):0.06):0.02,((t8:0.08,t9:0.07):0.05,(t10:0.14,t11:0.09):0.03):0.04):0.06);";
  pll_utree_t * tree = pll_utree_parse_newick_string(nwk);
  if (!tree) { fprintf(stderr, "newick: %s\n", pll_errmsg); return 1; }

  pllmod_treeinfo_t * ti = pllmod_treeinfo_create(tree->vroot, TAXA, 2, PLLMOD_COMMON_BRLEN_LINKED);
  if (!ti) { fprintf(stderr, "treeinfo: %s\n", pll_errmsg); return 1; }
  unsigned int params_indices[4] = {0, 0, 0, 0};
  pll_partition_t * parts[2];
  parts[0] = make_partition(tree, 4, 400, attrs);
  parts[1] = make_partition(tree, 5, 150, attrs);
  int sym4[6] = {0, 1, 2, 3, 4, 5}, sym5[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  if (!pllmod_treeinfo_init_partition(ti, 0, parts[0], PLLMOD_OPT_PARAM_BRANCHES_ITERATIVE,
                                      PLL_GAMMA_RATES_MEAN, 0.7, params_indices, sym4) ||
      !pllmod_treeinfo_init_partition(ti, 1, parts[1], PLLMOD_OPT_PARAM_BRANCHES_ITERATIVE,
                                      PLL_GAMMA_RATES_MEAN, 0.7, params_indices, sym5))
  { fprintf(stderr, "init_partition: %s\n", pll_errmsg); return 1; }

  printf("direct lnL:          %.6f\n", direct_lnl(tree, parts, 2));

  /* this repository's own driver (include/pllhip_eval.h) on a clone of the tree,
     same partitions, same optimiser settings as the reference call below */
  {
    pll_utree_t * copy = pll_utree_clone(tree);
    pllhip_eval_t * ev = pllhip_eval_create(copy, 2, 0);
    if (!ev || !pllhip_eval_set_partition(ev, 0, parts[0], params_indices) ||
        !pllhip_eval_set_partition(ev, 1, parts[1], params_indices))
    { fprintf(stderr, "eval: %s\n", pll_errmsg); return 1; }
    printf("driver full lnL:     %.6f\n", pllhip_eval_loglh(ev, 0));
    pllhip_eval_invalidate_clv(ev, pllhip_eval_root(ev)->next->back);
    printf("driver incremental:  %.6f\n", pllhip_eval_loglh(ev, 1));
    double l = -pllhip_eval_optimize_branches(ev, 1e-4, 10.0, 0.01, 8, PLLHIP_EVAL_RADIUS_ALL);
    if (pll_errno) { fprintf(stderr, "driver BLO: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
    printf("driver after BLO:    %.6f\n", l);
    pllhip_eval_destroy(ev);
    pll_utree_destroy(copy, NULL);
  }
  double l_full = pllmod_treeinfo_compute_loglh(ti, 0);
  printf("full lnL:            %.6f\n", l_full);

  pllmod_treeinfo_invalidate_clv(ti, ti->root->next->back);
  double l_inc = pllmod_treeinfo_compute_loglh(ti, 1);
  printf("incremental lnL:     %.6f\n", l_inc);

  double l_blo = -pllmod_algo_opt_brlen_treeinfo(ti, 1e-4, 10.0, 0.01, 8,
                                                 PLLMOD_OPT_BLO_NEWTON_FAST, PLLMOD_OPT_BRLEN_OPTIMIZE_ALL);
  if (pll_errno) { fprintf(stderr, "BLO: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
  printf("after BLO:           %.6f\n", l_blo);

  cutoff_info_t cutoff;
  memset(&cutoff, 0, sizeof(cutoff));
  double l_spr = pllmod_algo_spr_round(ti, 1, 5, 3, PLL_TRUE, PLLMOD_OPT_BLO_NEWTON_FAST,
                                       1e-4, 10.0, 4, 0.1, NULL, 0.0, 0.1, PLL_TRUE);
  if (pll_errno) { fprintf(stderr, "SPR: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
  printf("after SPR round:     %.6f\n", l_spr);

  double l_check = pllmod_treeinfo_compute_loglh(ti, 0);
  printf("full recomputation:  %.6f\n", l_check);

  pllmod_treeinfo_destroy(ti);
  pll_partition_destroy(parts[0]);
  pll_partition_destroy(parts[1]);
  pll_utree_destroy(tree, NULL);
  return 0;
}
