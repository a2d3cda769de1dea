/*
 * pll_notimpl.c -- tier B3 of include/pll.h: libpll-2 entry points that
 * pll-modules' NON-hot-path files (rooted trees, parsimony, sequence file
 * readers, PRNG) reference.  They are outside the scope of this engine
 * ("B3 out of scope") and exist only so that a program
 * that also links those files resolves its symbols.  Every one of them fails
 * loudly: it sets pll_errno = PLL_ERROR_NOT_IMPLEMENTED and returns
 * NULL / PLL_FAILURE.  Nothing on the likelihood path calls them.
 * (The SPR / NNI topology primitives that pllmod_algo_spr_round needs are real:
 * pll_utree_moves.c.)
 */
#include "pll.h"

static void notimpl(const char * what)
{
  pll_errno = PLL_ERROR_NOT_IMPLEMENTED;
  snprintf(pll_errmsg, 200, "%s is not implemented by the HIP engine (out of scope)", what);
}

#define NI_PTR(T, name, args) T name args { notimpl(#name); return NULL; }
#define NI_INT(name, args) int name args { notimpl(#name); return PLL_FAILURE; }
#define NI_VOID(name, args) void name args { notimpl(#name); }

NI_PTR(pll_rtree_t *, pll_rtree_parse_newick, (const char * f))
NI_VOID(pll_rtree_destroy, (pll_rtree_t * t, void (*cb)(void *)))
NI_PTR(char *, pll_rtree_export_newick, (const pll_rnode_t * r, char * (*cb)(const pll_rnode_t *)))
NI_VOID(pll_rtree_show_ascii, (const pll_rnode_t * t, int o))
NI_INT(pll_rtree_traverse, (pll_rnode_t * r, int t, int (*cb)(pll_rnode_t *), pll_rnode_t ** o, unsigned int * n))
NI_VOID(pll_rtree_create_operations, (pll_rnode_t * const * b, unsigned int n, double * br, unsigned int * pm, pll_operation_t * ops, unsigned int * mc, unsigned int * oc))
NI_PTR(pll_rtree_t *, pll_rtree_wraptree, (pll_rnode_t * r, unsigned int t))

NI_PTR(pll_parsimony_t *, pll_fastparsimony_init, (const pll_partition_t * p))
NI_VOID(pll_parsimony_destroy, (pll_parsimony_t * p))
NI_PTR(pll_utree_t *, pll_fastparsimony_stepwise, (pll_parsimony_t ** l, char * const * lab, unsigned int * s, unsigned int c, unsigned int seed))
NI_INT(pll_fastparsimony_stepwise_extend, (pll_utree_t * t, pll_parsimony_t ** l, unsigned int c, char * const * lab, unsigned int * m, unsigned int seed, unsigned int * s))
NI_INT(pll_fastparsimony_stepwise_spr_round, (pll_utree_t * t, pll_parsimony_t ** l, unsigned int c, const unsigned int * m, unsigned int seed, const int * v, unsigned int * cost))

NI_PTR(pll_fasta_t *, pll_fasta_open, (const char * f, const unsigned int * m))
NI_INT(pll_fasta_getnext, (pll_fasta_t * fd, char ** h, long * hl, char ** s, long * sl, long * no))
NI_VOID(pll_fasta_close, (pll_fasta_t * fd))
NI_INT(pll_fasta_rewind, (pll_fasta_t * fd))
NI_PTR(pll_msa_t *, pll_phylip_load, (const char * f, pll_bool_t i))
NI_VOID(pll_msa_destroy, (pll_msa_t * m))
NI_PTR(unsigned int *, pll_compress_site_patterns, (char ** s, const pll_state_t * m, int c, int * l))

