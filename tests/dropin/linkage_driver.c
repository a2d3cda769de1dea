/*
 * linkage_driver.c -- scaled and unlinked branch-length modes: pll-modules' treeinfo +
 * pllmod_algo_opt_brlen_treeinfo (compiled unchanged from /root/reference by
 * tests/test_dropin_modules.py) against this repository's evaluation driver
 * (include/pllhip_eval.h) on the same partitions, tree and settings.
 * Only PUBLIC pll-modules functions and struct fields are used.
 *   argv[1] = "scaled" | "unlinked"
 */
#include "pllmod_common.h"
#include "pll_tree.h"
#include "pll_optimize.h"
#include "pllmod_algorithm.h"
#include "pllhip_eval.h"
#include <stdio.h>

#define TAXA 10
#define PARTS 3

pll_split_t * pll_utree_split_newick_string(char * s, unsigned int tip_count, string_hashtable_t * names_hash)
{
  (void)s; (void)tip_count; (void)names_hash;
  abort();      /* flex/bison parser of pll-modules: not built, not reached (see treeinfo_driver.c) */
}

static unsigned long long rng_state = 0x7654321ULL;
static unsigned int rnd(unsigned int n)
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (unsigned int)((rng_state >> 33) % n);
}

static pll_partition_t * make_partition(const pll_utree_t * tree, unsigned int states, unsigned int sites,
                                        unsigned int change)
{
  unsigned int i, s;
  pll_partition_t * p = pll_partition_create(TAXA, TAXA - 2, states, sites, 1, 2 * TAXA - 3, 4, TAXA - 2,
                                             PLL_ATTRIB_PATTERN_TIP);
  if (!p) { fprintf(stderr, "partition: %s\n", pll_errmsg); exit(1); }
  double freqs[5], rates[10], cats[4];
  for (i = 0; i < states; ++i) freqs[i] = 1.0 / states;
  for (i = 0; i < states * (states - 1) / 2; ++i) rates[i] = 0.6 + 0.2 * (i % 4);
  rates[states * (states - 1) / 2 - 1] = 1.0;
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, rates);
  pll_compute_gamma_cats(0.9, 4, cats, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, cats);
  pll_state_t map[256];
  memset(map, 0, sizeof(map));
  for (s = 0; s < states; ++s) map['a' + s] = 1ULL << s;
  char * seq[TAXA];
  for (i = 0; i < TAXA; ++i) seq[i] = (char *)malloc(sites + 1);
  for (s = 0; s < sites; ++s)
  {
    char base = (char)('a' + rnd(states));
    /* partitions evolve at different speeds: what scalers / unlinked lengths are for */
    for (i = 0; i < TAXA; ++i) seq[i][s] = (rnd(100) < change) ? (char)('a' + rnd(states)) : base;
  }
  for (i = 0; i < TAXA; ++i)
  {
    seq[i][sites] = 0;
    if (!pll_set_tip_states(p, tree->nodes[i]->clv_index, map, seq[i])) exit(1);
    free(seq[i]);
  }
  return p;
}

int main(int argc, char ** argv)
{
  const int unlinked = argc > 1 && !strcmp(argv[1], "unlinked");
  const int linkage = unlinked ? PLLMOD_COMMON_BRLEN_UNLINKED : PLLMOD_COMMON_BRLEN_SCALED;
  const char * nwk = "((t0:0.11,t1:0.07):0.05,(t2:0.13,(t3:0.06,t4:0.09):0.04):0.03,"
                     "((t5:0.10,(t6:0.05,t7:0.12):0.06):0.02,(t8:0.08,t9:0.07):0.05):0.06);";
  const double scalers[PARTS] = {0.6, 1.0, 2.2};
  pll_utree_t * tree = pll_utree_parse_newick_string(nwk);
  if (!tree) { fprintf(stderr, "newick: %s\n", pll_errmsg); return 1; }
  unsigned int params_indices[4] = {0, 0, 0, 0}, p, i, m;
  const unsigned int nodes = tree->tip_count + tree->inner_count, edges = tree->edge_count;
  pll_partition_t * parts[PARTS];
  parts[0] = make_partition(tree, 4, 500, 10);
  parts[1] = make_partition(tree, 4, 300, 25);
  parts[2] = make_partition(tree, 5, 200, 45);

  /* --- this repository's driver, on a clone of the tree ------------------------------ */
  {
    pll_utree_t * copy = pll_utree_clone(tree);
    pllhip_eval_t * ev = pllhip_eval_create(copy, PARTS, 0);
    if (!ev) { fprintf(stderr, "eval: %s\n", pll_errmsg); return 1; }
    for (p = 0; p < PARTS; ++p) if (!pllhip_eval_set_partition(ev, p, parts[p], params_indices)) return 1;
    if (!pllhip_eval_set_brlen_linkage(ev, unlinked ? PLLHIP_EVAL_BRLEN_UNLINKED : PLLHIP_EVAL_BRLEN_SCALED)) return 1;
    for (p = 0; p < PARTS; ++p)
    {
      if (!unlinked) { if (!pllhip_eval_set_brlen_scaler(ev, p, scalers[p])) return 1; continue; }
      for (i = 0; i < nodes; ++i)
      {
        pll_unode_t * s = copy->nodes[i];
        do
        {
          if (!pllhip_eval_set_partition_branch_length(ev, p, s, s->length * scalers[p])) return 1;
          s = s->next;
        } while (s && s != copy->nodes[i]);
      }
    }
    printf("driver lnL:        %.6f\n", pllhip_eval_loglh(ev, 0));
    double l = -pllhip_eval_optimize_branches(ev, 1e-4, 10.0, 0.01, 8, PLLHIP_EVAL_RADIUS_ALL);
    if (pll_errno) { fprintf(stderr, "driver BLO: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
    printf("driver after BLO:  %.6f\n", l);
    printf("driver re-eval:    %.6f\n", pllhip_eval_loglh(ev, 0));
    for (p = 0; p < PARTS; ++p)
    {
      double sum = 0.0, one = 0.0;
      for (i = 0; i < nodes; ++i)
      {
        pll_unode_t * s = copy->nodes[i];
        do
        {
          const double t = pllhip_eval_get_partition_branch_length(ev, p, s) * pllhip_eval_get_brlen_scaler(ev, p);
          sum += t / 2;                  /* every branch is seen from both ends */
          if (s->pmatrix_index == 5) one = t;
          s = s->next;
        } while (s && s != copy->nodes[i]);
      }
      printf("driver tree length %u: %.6f\n", p, sum);
      printf("driver branch 5 of %u: %.6f\n", p, one);
    }
    pllhip_eval_destroy(ev);
    pll_utree_destroy(copy, NULL);
  }

  /* --- the reference ----------------------------------------------------------------- */
  pllmod_treeinfo_t * ti = pllmod_treeinfo_create(tree->vroot, TAXA, PARTS, linkage);
  if (!ti) { fprintf(stderr, "treeinfo: %s\n", pll_errmsg); return 1; }
  int sym4[6] = {0, 1, 2, 3, 4, 5}, sym5[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  for (p = 0; p < PARTS; ++p)
    if (!pllmod_treeinfo_init_partition(ti, p, parts[p], PLLMOD_OPT_PARAM_BRANCHES_ITERATIVE,
                                        PLL_GAMMA_RATES_MEAN, 0.9, params_indices, p == 2 ? sym5 : sym4))
    { fprintf(stderr, "init_partition: %s\n", pll_errmsg); return 1; }
  for (p = 0; p < PARTS; ++p)
  {
    if (!unlinked) { ti->brlen_scalers[p] = scalers[p]; continue; }
    for (m = 0; m < edges; ++m) ti->branch_lengths[p][m] *= scalers[p];
  }
  pllmod_treeinfo_invalidate_all(ti);
  printf("lnL:               %.6f\n", pllmod_treeinfo_compute_loglh(ti, 0));
  double l_blo = -pllmod_algo_opt_brlen_treeinfo(ti, 1e-4, 10.0, 0.01, 8, PLLMOD_OPT_BLO_NEWTON_FAST,
                                                 PLLMOD_OPT_BRLEN_OPTIMIZE_ALL);
  if (pll_errno) { fprintf(stderr, "BLO: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
  printf("after BLO:         %.6f\n", l_blo);
  printf("re-eval:           %.6f\n", pllmod_treeinfo_compute_loglh(ti, 0));
  for (p = 0; p < PARTS; ++p)
  {
    double sum = 0.0;
    const double s = unlinked ? 1.0 : ti->brlen_scalers[p];
    for (m = 0; m < edges; ++m) sum += ti->branch_lengths[p][m] * s;
    printf("tree length %u:     %.6f\n", p, sum);
    printf("branch 5 of %u:     %.6f\n", p, ti->branch_lengths[p][5] * s);
  }
  pllmod_treeinfo_destroy(ti);
  for (p = 0; p < PARTS; ++p) pll_partition_destroy(parts[p]);
  pll_utree_destroy(tree, NULL);
  return 0;
}
