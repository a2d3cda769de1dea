/*
 * binary_driver.c -- the reference's binary (checkpoint) module, compiled unchanged from
 * /root/reference/src/binary by tests/test_dropin_modules.py, on this repository's
 * library: dump a partition with its CLVs, scalers and P-matrices, load it into a new
 * partition, and evaluate the edge log-likelihood WITHOUT recomputing anything
 * (the reference's own criterion, test/src/binary/binary-sequential.c:344-354).
 * usage: binary_driver <file> tv|clv|tv-repeats|clv-repeats
 * "-repeats": the partition is created with PLL_ATTRIB_SITE_REPEATS; the reference then dereferences
 * partition->repeats in its dump / load walk (src/binary/binary_io_operations.c:231-236, 265-282;
 * src/binary/pll_binary.c:388-406) and in pllmod_msa_empirical_frequencies (src/msa/pll_msa.c:108-112),
 * which is called here on every partition (the reference's own src/msa/pll_msa.c, compiled unchanged).
 */
#include "pllmod_common.h"
#include "pll_binary.h"
#include "pll_msa.h"
#include <stdio.h>

#define TAXA 9
#define BLOCK_ID_PARTITION 3

static unsigned long long rng_state = 99;
static unsigned int rnd(unsigned int n)
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (unsigned int)((rng_state >> 33) % n);
}

static int cb_all(pll_unode_t * n) { (void)n; return 1; }

int main(int argc, char ** argv)
{
  if (argc < 3) return 2;
  const unsigned int attrs = (!strncmp(argv[2], "tv", 2) ? PLL_ATTRIB_PATTERN_TIP : 0) |
                             (strstr(argv[2], "-repeats") ? PLL_ATTRIB_SITE_REPEATS : 0);
  const unsigned int states = 20, sites = 333, cats = 4;
  const char * nwk = "((t0:0.11,t1:0.07):0.05,(t2:0.13,(t3:0.06,t4:0.09):0.04):0.03,"
                     "((t5:0.10,t6:0.05):0.02,(t7:0.08,t8:0.07):0.05):0.06);";
  pll_utree_t * tree = pll_utree_parse_newick_string(nwk);
  if (!tree) { fprintf(stderr, "newick: %s\n", pll_errmsg); return 1; }
  pll_partition_t * p = pll_partition_create(TAXA, TAXA - 2, states, sites, 1, 2 * TAXA - 3, cats,
                                             TAXA - 2, attrs);
  if (!p) { fprintf(stderr, "partition: %s\n", pll_errmsg); return 1; }
  unsigned int i, s;
  double freqs[20], rates[190], gamma[4];
  for (i = 0; i < states; ++i) freqs[i] = 1.0 / states;
  for (i = 0; i < 190; ++i) rates[i] = 0.5 + 0.1 * (i % 7);
  rates[189] = 1.0;
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, rates);
  pll_compute_gamma_cats(0.6, cats, gamma, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, gamma);
  pll_update_invariant_sites_proportion(p, 0, 0.0);
  unsigned int weights[333];
  for (s = 0; s < sites; ++s) weights[s] = 1 + rnd(3);
  pll_set_pattern_weights(p, weights);
  const char * aa = "ARNDCQEGHILKMFPSTWYV";
  char seq[334];
  for (i = 0; i < TAXA; ++i)
  {
    for (s = 0; s < sites; ++s) seq[s] = (rnd(25) == 0) ? '-' : aa[(s * 7 + rnd(4) + i / 3) % 20];
    seq[sites] = 0;
    if (!pll_set_tip_states(p, tree->nodes[i]->clv_index, pll_map_aa, seq))
    { fprintf(stderr, "tips: %s\n", pll_errmsg); return 1; }
  }

  unsigned int n = 0, nm = 0, no = 0, params[4] = {0, 0, 0, 0};
  pll_unode_t * trav[2 * TAXA];
  double brlens[2 * TAXA];
  unsigned int midx[2 * TAXA];
  pll_operation_t ops[TAXA];
  pll_unode_t * root = tree->vroot;
  if (!pll_utree_traverse(root, PLL_TREE_TRAVERSE_POSTORDER, cb_all, trav, &n)) return 1;
  pll_utree_create_operations(trav, n, brlens, midx, ops, &nm, &no);
  if (!pll_update_prob_matrices(p, params, midx, brlens, nm)) return 1;
  pll_update_partials(p, ops, no);
  const unsigned int pc = root->clv_index, cc = root->back->clv_index, pm = root->pmatrix_index;
  const int ps = root->scaler_index, cs = root->back->scaler_index;
  const double before = pll_compute_edge_loglikelihood(p, pc, ps, cc, cs, pm, params, NULL);
  printf("lnL before: %.10f\n", before);
  {
    double * ef = pllmod_msa_empirical_frequencies(p);
    if (!ef) { fprintf(stderr, "empirical frequencies: %s\n", pll_errmsg); return 1; }
    printf("freqs:");
    for (i = 0; i < states; ++i) printf(" %.12f", ef[i]);
    printf("\n");
    free(ef);
  }

  pll_binary_header_t header;
  FILE * f = pllmod_binary_create(argv[1], &header, PLLMOD_BIN_ACCESS_SEQUENTIAL, 0);
  if (!f) { fprintf(stderr, "create: %s\n", pll_errmsg); return 1; }
  if (!pllmod_binary_partition_dump(f, BLOCK_ID_PARTITION, p,
                                    PLLMOD_BIN_ATTRIB_PARTITION_DUMP_CLV | PLLMOD_BIN_ATTRIB_PARTITION_DUMP_WGT))
  { fprintf(stderr, "dump: %s\n", pll_errmsg); return 1; }
  pllmod_binary_close(f);
  pll_partition_destroy(p);

  unsigned int battr = 0;
  f = pllmod_binary_open(argv[1], &header);
  if (!f) { fprintf(stderr, "open: %s\n", pll_errmsg); return 1; }
  p = pllmod_binary_partition_load(f, BLOCK_ID_PARTITION, NULL, &battr, 0);
  if (!p) { fprintf(stderr, "load: %s\n", pll_errmsg); return 1; }
  pllmod_binary_close(f);
  const double after = pll_compute_edge_loglikelihood(p, pc, ps, cc, cs, pm, params, NULL);
  printf("lnL after:  %.10f\n", after);
  printf("weights: %u\n", p->pattern_weight_sum);
  printf("attributes restored: %s\n", ((p->attributes ^ attrs) & (PLL_ATTRIB_SITE_REPEATS | PLL_ATTRIB_PATTERN_TIP)) ? "no" : "yes");
  if ((attrs & PLL_ATTRIB_SITE_REPEATS) && !p->repeats) { fprintf(stderr, "loaded partition without repeats table\n"); return 1; }
  /* the restored partition keeps working: same result after recomputing everything */
  if (!pll_update_prob_matrices(p, params, midx, brlens, nm)) return 1;
  pll_update_partials(p, ops, no);
  printf("lnL redo:   %.10f\n", pll_compute_edge_loglikelihood(p, pc, ps, cc, cs, pm, params, NULL));
  pll_partition_destroy(p);
  pll_utree_destroy(tree, NULL);
  return 0;
}
