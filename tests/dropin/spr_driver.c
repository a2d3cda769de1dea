/*
 * spr_driver.c -- runs the reference's pllmod_algo_spr_round (compiled unchanged
 * from /root/reference by tests/test_dropin_modules.py) and this repository's
 * pllhip_eval_spr_round (include/pllhip_eval.h) on the same starting tree and
 * data, both on the library the test links (the CPU oracle build), and prints
 * what each one ends with.  Only public pll-modules / include/pll.h calls.
 *
 * usage: spr_driver <newick file> <alignment file> fast|thorough <radius_max> <rounds> <ntopol> [linked|scaled|unlinked]
 * alignment file: one "<label> <sequence over acgt->" line per taxon; columns are
 * split into two partitions (first 60 % / rest) with different models.
 */
#include "pllmod_common.h"
#include "pll_tree.h"
#include "pll_optimize.h"
#include "pllmod_algorithm.h"
#include "pllhip_eval.h"
#include <stdio.h>

/* see treeinfo_driver.c: the flex/bison split parser is not part of this build */
pll_split_t * pll_utree_split_newick_string(char * s, unsigned int tip_count,
                                            string_hashtable_t * names_hash)
{
  (void)s; (void)tip_count; (void)names_hash;
  abort();
}

#define MAXTAXA 64
static char * labels[MAXTAXA], * seqs[MAXTAXA];
static unsigned int ntaxa, nsites;

static void read_alignment(const char * path)
{
  FILE * f = fopen(path, "r");
  static char lab[256], seq[1 << 16];
  if (!f) { perror(path); exit(1); }
  while (ntaxa < MAXTAXA && fscanf(f, "%255s %65535s", lab, seq) == 2)
  {
    labels[ntaxa] = strdup(lab);
    seqs[ntaxa] = strdup(seq);
    nsites = (unsigned int)strlen(seq);
    ++ntaxa;
  }
  fclose(f);
}

static pll_partition_t * make_partition(const pll_utree_t * tree, unsigned int first, unsigned int count,
                                        double alpha, double kappa)
{
  unsigned int i, j;
  pll_partition_t * p = pll_partition_create(ntaxa, ntaxa - 2, 4, count, 1, 2 * ntaxa - 3, 4,
                                             ntaxa - 2, PLL_ATTRIB_PATTERN_TIP);
  if (!p) { fprintf(stderr, "partition: %s\n", pll_errmsg); exit(1); }
  double freqs[4] = {0.3, 0.2, 0.2, 0.3}, rates[6] = {1.0, kappa, 1.0, 1.0, kappa, 1.0}, cats[4];
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, rates);
  pll_compute_gamma_cats(alpha, 4, cats, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, cats);
  char * buf = (char *)malloc(count + 1);
  for (i = 0; i < ntaxa; ++i)
  {
    const pll_unode_t * tip = tree->nodes[i];
    for (j = 0; j < ntaxa; ++j) if (!strcmp(labels[j], tip->label)) break;
    if (j == ntaxa) { fprintf(stderr, "no sequence for %s\n", tip->label); exit(1); }
    memcpy(buf, seqs[j] + first, count);
    buf[count] = 0;
    if (!pll_set_tip_states(p, tip->clv_index, pll_map_nt, buf)) { fprintf(stderr, "tips: %s\n", pll_errmsg); exit(1); }
  }
  free(buf);
  return p;
}

static char * slurp(const char * path)
{
  FILE * f = fopen(path, "r");
  static char buf[1 << 16];
  if (!f) { perror(path); exit(1); }
  size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  buf[n] = 0;
  fclose(f);
  return buf;
}

int main(int argc, char ** argv)
{
  if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
  const int thorough = !strcmp(argv[3], "thorough");
  const unsigned int radius_max = (unsigned int)atoi(argv[4]);
  const int rounds = atoi(argv[5]);
  const unsigned int ntopol = (unsigned int)atoi(argv[6]);
  const char * linkage = argc > 7 ? argv[7] : "linked";
  const int ref_linkage = !strcmp(linkage, "unlinked") ? PLLMOD_COMMON_BRLEN_UNLINKED
                        : !strcmp(linkage, "scaled") ? PLLMOD_COMMON_BRLEN_SCALED : PLLMOD_COMMON_BRLEN_LINKED;
  const int own_linkage = !strcmp(linkage, "unlinked") ? PLLHIP_EVAL_BRLEN_UNLINKED
                        : !strcmp(linkage, "scaled") ? PLLHIP_EVAL_BRLEN_SCALED : PLLHIP_EVAL_BRLEN_LINKED;
  int round;
  read_alignment(argv[2]);
  pll_utree_t * tree = pll_utree_parse_newick_string(slurp(argv[1]));
  if (!tree || tree->tip_count != ntaxa) { fprintf(stderr, "newick: %s\n", pll_errmsg); return 1; }
  pll_utree_t * copy = pll_utree_clone(tree);

  const unsigned int cut = nsites * 6 / 10;
  unsigned int params_indices[4] = {0, 0, 0, 0};
  int sym[6] = {0, 1, 2, 3, 4, 5};
  pll_partition_t * parts[2];
  parts[0] = make_partition(tree, 0, cut, 0.6, 2.5);
  parts[1] = make_partition(tree, cut, nsites - cut, 1.1, 4.0);

  /* ---- the reference ---- */
  pllmod_treeinfo_t * ti = pllmod_treeinfo_create(tree->vroot, ntaxa, 2, ref_linkage);
  if (!ti) { fprintf(stderr, "treeinfo: %s\n", pll_errmsg); return 1; }
  if (!pllmod_treeinfo_init_partition(ti, 0, parts[0], PLLMOD_OPT_PARAM_BRANCHES_ITERATIVE,
                                      PLL_GAMMA_RATES_MEAN, 0.6, params_indices, sym) ||
      !pllmod_treeinfo_init_partition(ti, 1, parts[1], PLLMOD_OPT_PARAM_BRANCHES_ITERATIVE,
                                      PLL_GAMMA_RATES_MEAN, 1.1, params_indices, sym))
  { fprintf(stderr, "init_partition: %s\n", pll_errmsg); return 1; }
  if (ref_linkage == PLLMOD_COMMON_BRLEN_SCALED) { ti->brlen_scalers[0] = 0.8; ti->brlen_scalers[1] = 1.3; }
  cutoff_info_t rc;
  memset(&rc, 0, sizeof(rc));
  rc.lh_cutoff = 1e30;       /* first round: no cutoff yet, as raxml-ng starts it */
  for (round = 0; round < rounds; ++round)
  {
    double l = pllmod_algo_spr_round(ti, 1, radius_max, ntopol, thorough, PLLMOD_OPT_BLO_NEWTON_FAST,
                                     1e-4, 10.0, 8, 0.1, &rc, 1.0, 0.1, PLL_TRUE);
    if (!l || pll_errno) { fprintf(stderr, "reference SPR: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
    printf("ref round %d lnL: %.8f\n", round, l);
    printf("ref round %d cutoff: %d %.8f %.8f\n", round, rc.lh_dec_count, rc.lh_dec_sum, rc.lh_cutoff);
  }
  char * nw = pll_utree_export_newick(ti->root, NULL);
  printf("ref tree: %s\n", nw);
  free(nw);
  {
    /* per-partition tree lengths (the newick lengths mean nothing with unlinked branch lengths) */
    unsigned int p, k;
    for (p = 0; p < 2; ++p)
    {
      double tot = 0.0;
      const unsigned int src = (ref_linkage == PLLMOD_COMMON_BRLEN_UNLINKED) ? p : 0;
      for (k = 0; k < 2 * ntaxa - 3; ++k) tot += ti->branch_lengths[src][k];
      printf("ref plen %u: %.8f\n", p, tot);
    }
  }
  pllmod_treeinfo_destroy(ti);

  /* ---- this repository's round, from the same start ---- */
  pllhip_eval_t * ev = pllhip_eval_create(copy, 2, 0);
  if (!ev || !pllhip_eval_set_partition(ev, 0, parts[0], params_indices) ||
      !pllhip_eval_set_partition(ev, 1, parts[1], params_indices))
  { fprintf(stderr, "eval: %s\n", pll_errmsg); return 1; }
  if (!pllhip_eval_set_brlen_linkage(ev, own_linkage)) { fprintf(stderr, "linkage: %s\n", pll_errmsg); return 1; }
  if (own_linkage == PLLHIP_EVAL_BRLEN_SCALED)
  { pllhip_eval_set_brlen_scaler(ev, 0, 0.8); pllhip_eval_set_brlen_scaler(ev, 1, 1.3); }
  pllhip_spr_params_t prm;
  memset(&prm, 0, sizeof(prm));
  prm.radius_min = 1; prm.radius_max = radius_max; prm.ntopol_keep = ntopol; prm.thorough = thorough;
  prm.bl_min = 1e-4; prm.bl_max = 10.0; prm.smoothings = 8; prm.epsilon = 0.1;
  prm.subtree_cutoff = 1.0; prm.lh_epsilon_brlen_triplet = 0.1;
  pllhip_spr_cutoff_t mc;
  memset(&mc, 0, sizeof(mc));
  mc.lh_cutoff = 1e30;
  for (round = 0; round < rounds; ++round)
  {
    pllhip_spr_stats_t st;
    double l = pllhip_eval_spr_round(ev, &prm, &mc, &st);
    if (!l) { fprintf(stderr, "own SPR: [%d] %s\n", pll_errno, pll_errmsg); return 1; }
    printf("own round %d lnL: %.8f\n", round, l);
    printf("own round %d cutoff: %d %.8f %.8f\n", round, mc.lh_dec_count, mc.lh_dec_sum, mc.lh_cutoff);
    printf("own round %d stats: prunings %lu insertions %lu applied %lu rescored %lu\n", round,
           st.prunings, st.insertions, st.moves_applied, st.rescored);
  }
  nw = pll_utree_export_newick(pllhip_eval_root(ev), NULL);
  printf("own tree: %s\n", nw);
  free(nw);
  {
    unsigned int p, k;
    for (p = 0; p < 2; ++p)
    {
      double tot = 0.0;
      for (k = 0; k < copy->tip_count + copy->inner_count; ++k)
      {
        pll_unode_t * n = copy->nodes[k], * t = n;
        do
        {
          if (t->node_index < t->back->node_index) tot += pllhip_eval_get_partition_branch_length(ev, p, t);
          t = t->next;
        } while (t && t != n);
      }
      printf("own plen %u: %.8f\n", p, tot);
    }
  }
  pllhip_eval_destroy(ev);
  pll_utree_destroy(copy, NULL);
  pll_partition_destroy(parts[0]);
  pll_partition_destroy(parts[1]);
  pll_utree_destroy(tree, NULL);
  return 0;
}
