/*
 * pll.h -- C ABI of the MI355X-native likelihood engine ("pll-hip").
 *
 * This header is the drop-in boundary of the project.  It declares, with the
 * names, argument order and field names that ddarriba/pll-modules dereferences,
 * the subset of the libpll-2 interface that pll-modules' tree / optimize /
 * algorithm layers call.  libpll-2 itself is NOT available to this project
 * (the reference checkout has an empty libs/libpll), so
 * every declaration below was reconstructed from the *call sites* in the
 * reference tree; each block cites the file:line that fixes its shape.
 *
 * Two libraries export this interface:
 *   - pll-modules_amd/libpll_hip.so : the product (HIP kernels for gfx950).
 *   - oracle/_build/libpll_oracle.so: plain-C CPU restatement (test
 *     infrastructure only; see oracle/README.md).
 *
 * Tiers:
 *   B0 kernels            pll_update_partials, pll_update_prob_matrices,
 *                         pll_compute_edge_loglikelihood,
 *                         pll_compute_root_loglikelihood, pll_update_sumtable,
 *                         pll_compute_likelihood_derivatives
 *   B1 lifecycle/setters  pll_partition_create/destroy, pll_set_*,
 *                         pll_compute_gamma_cats, pll_aligned_alloc/free,
 *                         pll_errno / pll_errmsg
 *   B2 tree utilities     pll_unode_t, pll_utree_t, traverse,
 *                         create_operations, wraptree, clone, destroy
 */
#ifndef PLLHIP_PLL_H_INCLUDED
#define PLLHIP_PLL_H_INCLUDED

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLL_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* constants                                                          */
/* ------------------------------------------------------------------ */

/* return convention: reference checks `if (!pll_update_prob_matrices(...))`
   (src/tree/treeinfo.c:854-861) */
#define PLL_FAILURE 0
#define PLL_SUCCESS 1

#define PLL_FALSE 0
#define PLL_TRUE  1

#define PLL_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define PLL_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define PLL_SWAP(x, y) do { __typeof__(x) SWAP_ = (x); (x) = (y); (y) = SWAP_; } while (0)

#define PLL_ONE_EPSILON 1e-15
#define PLL_ASCII_SIZE  256

/* memory alignment handed to pll_aligned_alloc by callers
   (src/tree/treeinfo.c:339, src/optimize/pll_optimize.c:1024) */
#define PLL_ALIGNMENT_CPU 8
#define PLL_ALIGNMENT_SSE 16
#define PLL_ALIGNMENT_AVX 32
#define PLL_ALIGNMENT_HIP 64

/* attribute bits (test/src/common.c:10-32, src/tree/treeinfo.c:334).
   The ARCH bits select a CPU kernel family in libpll; this engine has one
   kernel family (HIP), so ARCH bits are accepted and ignored. */
#define PLL_ATTRIB_ARCH_CPU       0
#define PLL_ATTRIB_ARCH_SSE       (1 << 0)
#define PLL_ATTRIB_ARCH_AVX       (1 << 1)
#define PLL_ATTRIB_ARCH_AVX2      (1 << 2)
#define PLL_ATTRIB_ARCH_AVX512    (1 << 3)
#define PLL_ATTRIB_ARCH_MASK      0xF
#define PLL_ATTRIB_PATTERN_TIP    (1 << 4)
#define PLL_ATTRIB_AB_LEWIS       (1 << 5)
#define PLL_ATTRIB_AB_FELSENSTEIN (2 << 5)
#define PLL_ATTRIB_AB_STAMATAKIS  (3 << 5)
#define PLL_ATTRIB_AB_MASK        (7 << 5)
#define PLL_ATTRIB_AB_FLAG        (1 << 8)
#define PLL_ATTRIB_RATE_SCALERS   (1 << 9)
#define PLL_ATTRIB_SITE_REPEATS   (1 << 10)
#define PLL_ATTRIB_MASK           ((1 << 11) - 1)

/* per-site numerical scaling: a site whose R*S entries all fall below
   2^-256 is multiplied by 2^256 and its scaler count incremented */
#define PLL_SCALE_FACTOR      115792089237316195423570985008687907853269984665640564039457584007913129639936.0 /* 2^256 */
#define PLL_SCALE_THRESHOLD   (1.0 / PLL_SCALE_FACTOR)
#define PLL_SCALE_BUFFER_NONE (-1)

#define PLL_MISC_EPSILON 1e-8

/* discrete gamma modes (src/tree/treeinfo.c:174) */
#define PLL_GAMMA_RATES_MEAN   0
#define PLL_GAMMA_RATES_MEDIAN 1

/* error codes: libpll codes are < 1000, pll-modules uses 1001..5999
   (src/pllmod_common.h:37-41) */
#define PLL_ERROR_FILE_OPEN        100
#define PLL_ERROR_FILE_SEEK        101
#define PLL_ERROR_FILE_EOF         102
#define PLL_ERROR_FASTA_ILLEGALCHAR   201
#define PLL_ERROR_FASTA_UNPRINTABLECHAR 202
#define PLL_ERROR_FASTA_INVALIDHEADER 203
#define PLL_ERROR_FASTA_NONALIGNED    204
#define PLL_ERROR_PHYLIP_SYNTAX       231
#define PLL_ERROR_PHYLIP_LONGSEQ      232
#define PLL_ERROR_PHYLIP_NONALIGNED   233
#define PLL_ERROR_PHYLIP_ILLEGALCHAR  234
#define PLL_ERROR_PHYLIP_UNPRINTABLECHAR 235
#define PLL_ERROR_NEWICK_SYNTAX    111
#define PLL_ERROR_MEM_ALLOC        112
#define PLL_ERROR_PARAM_INVALID    113
#define PLL_ERROR_TIPDATA_ILLEGALSTATE 114
#define PLL_ERROR_TIPDATA_ILLEGALFUNCTION 115
#define PLL_ERROR_TREE_CONVERSION  116
#define PLL_ERROR_INVAR_INCOMPAT   117
#define PLL_ERROR_INVAR_PROPORTION 118
#define PLL_ERROR_INVAR_PARAMINDEX 119
#define PLL_ERROR_INVAR_NONEFOUND  120
#define PLL_ERROR_AB_INVALIDMETHOD 121
#define PLL_ERROR_AB_NOSUPPORT     122
#define PLL_ERROR_SPR_TERMINALBRANCH 123
#define PLL_ERROR_SPR_NOCHANGE     124
#define PLL_ERROR_NNI_INVALIDMOVE  125
#define PLL_ERROR_NNI_TERMINALBRANCH 126
#define PLL_ERROR_STEPWISE_STRUCT  127
#define PLL_ERROR_STEPWISE_TIPS    128
#define PLL_ERROR_STEPWISE_UNSUPPORTED 129
#define PLL_ERROR_EINVAL           130
#define PLL_ERROR_MSA_EMPTY        131
#define PLL_ERROR_MSA_MAP_INVALID  132
#define PLL_ERROR_TREE_INVALID     133
/* engine-specific codes (kept below 1000 like every libpll-level code) */
#define PLL_ERROR_HIP_RUNTIME      900 /* a HIP / RCCL call failed        */
#define PLL_ERROR_HIP_NODEVICE     901 /* no gfx950 device visible        */
#define PLL_ERROR_NOT_IMPLEMENTED  902 /* declared for link compatibility */
#define PLL_ERROR_HIP_TIMEOUT      903 /* a result that depends on another rank did not arrive (peer lost) */
#define PLL_ERROR_HIP_COMM_ABORTED 904 /* the communicator was aborted after a failure; no further collectives */

/* tree traversal orders */
#define PLL_TREE_TRAVERSE_POSTORDER 1
#define PLL_TREE_TRAVERSE_PREORDER  2

/* utree printing options (src/algorithm/algo_search.c:759) */
#define PLL_UTREE_SHOW_LABEL          (1 << 0)
#define PLL_UTREE_SHOW_BRANCH_LENGTH  (1 << 1)
#define PLL_UTREE_SHOW_CLV_INDEX      (1 << 2)
#define PLL_UTREE_SHOW_SCALER_INDEX   (1 << 3)
#define PLL_UTREE_SHOW_PMATRIX_INDEX  (1 << 4)
#define PLL_UTREE_SHOW_DATA           (1 << 5)

/* topological moves (src/tree/pll_tree.c:72-293) */
#define PLL_UTREE_MOVE_SPR        0
#define PLL_UTREE_MOVE_NNI        1
#define PLL_UTREE_MOVE_NNI_LEFT   1
#define PLL_UTREE_MOVE_NNI_RIGHT  2
#define PLL_NNI_LEFT              1
#define PLL_NNI_RIGHT             2

/* popcount / ctz on state masks (src/msa/pll_msa.c, src/tree) */
#define PLL_POPCNT32      __builtin_popcount
#define PLL_POPCNT64      __builtin_popcountll
#define PLL_CTZ32         __builtin_ctz
#define PLL_CTZ64         __builtin_ctzll
#define PLL_STATE_POPCNT  __builtin_popcountll
#define PLL_STATE_CTZ     __builtin_ctzll

/* ------------------------------------------------------------------ */
/* types                                                              */
/* ------------------------------------------------------------------ */

/* state bitmask: bit i set <=> state i compatible (up to 64 states;
   test/src/optimize/blopt-5states.c:28-39 declares maps of this type) */
typedef unsigned long long pll_state_t;
typedef int pll_bool_t;

/* site-repeats bookkeeping: the fields pll-modules touches (src/binary/binary_io_operations.c:231-236,
   265-282, 329-390; src/binary/pll_binary.c:388-406, 655, 747, 841-851; src/msa/pll_msa.c:108-112).
   A partition created with PLL_ATTRIB_SITE_REPEATS carries a table that says "no node is compressed"
   (pernode_ids[i] = 0, perscale_ids[i] = 0, pernode_allocated_clvs[i] = pll_get_sites_number(), per-node
   index arrays NULL; csrc/host/pll_repeats.c): the engine computes per class of sites on the device
   (4 and 20 states, 2 .. 28 states with four rate categories; per-site scalers; with or without
   PLL_ATTRIB_PATTERN_TIP; other partitions ignore the attribute), but every
   vector a caller can see -- host mirrors, pllhip_get_clv, a checkpoint -- is site-indexed. */
struct pll_partition;
typedef struct pll_repeats
{
  unsigned int ** pernode_site_id;
  unsigned int ** pernode_id_site;
  unsigned int * pernode_ids;
  unsigned int * perscale_ids;
  unsigned int * pernode_allocated_clvs;
  unsigned int (*enable_repeats)(struct pll_partition *, unsigned int, unsigned int);
  void (*reallocate_repeats)(struct pll_partition *, unsigned int, int, unsigned int);
} pll_repeats_t;

/* The partition: field names exactly as dereferenced by pll-modules
   (src/binary/binary_io_operations.c:169-311, src/tree/treeinfo.c:292-340,
   src/optimize/pll_optimize.c:85-102, src/msa/pll_msa.c:48-124).
   Host arrays `rates, rate_weights, subst_params[i], frequencies[i],
   prop_invar, eigen_decomp_valid` are the source of truth: callers write them
   directly and every kernel-entry call re-syncs them.
   In the HIP build `clv[i]` / `scale_buffer[i]` are NULL until materialised
   with pllhip_sync_to_host() (include/pllhip.h): the data lives in HBM. */
typedef struct pll_partition
{
  unsigned int tips;
  unsigned int clv_buffers;
  unsigned int nodes;             /* tips + clv_buffers */
  unsigned int states;
  unsigned int sites;
  unsigned int pattern_weight_sum;
  unsigned int rate_matrices;
  unsigned int prob_matrices;
  unsigned int rate_cats;
  unsigned int scale_buffers;
  unsigned int attributes;

  size_t alignment;
  unsigned int states_padded;

  double ** clv;                  /* [nodes]        N*R*Sp, states fastest  */
  double ** pmatrix;              /* [prob_matrices] R*S*Sp; pmatrix[0] is
                                     one contiguous block                   */
  double * rates;                 /* [R]                                    */
  double * rate_weights;          /* [R]                                    */
  double ** subst_params;         /* [rate_matrices] S(S-1)/2               */
  unsigned int ** scale_buffer;   /* [scale_buffers] N                      */
  double ** frequencies;          /* [rate_matrices] Sp                     */
  double * prop_invar;            /* [rate_matrices]                        */
  int * invariant;                /* [N] state index or -1                  */
  unsigned int * pattern_weights; /* [N]                                    */

  int * eigen_decomp_valid;       /* [rate_matrices]                        */
  double ** eigenvecs;            /* [rate_matrices] S*Sp                   */
  double ** inv_eigenvecs;        /* [rate_matrices] S*Sp                   */
  double ** eigenvals;            /* [rate_matrices] Sp                     */

  /* tip codes (PLL_ATTRIB_PATTERN_TIP) */
  unsigned int maxstates;         /* number of distinct tip codes           */
  unsigned char ** tipchars;      /* [tips] N codes                         */
  unsigned char * charmap;        /* [256] input char -> code               */
  double * ttlookup;              /* sized as libpll does, zero-filled: only the binary dump reads it */
  pll_state_t * tipmap;           /* [256] code -> state mask               */

  int asc_bias_alloc;
  int asc_additional_sites;

  struct pll_repeats * repeats;

  /* engine-private state (device buffers, streams, cached model state).
     Not part of the libpll layout; appended so that source-level users are
     unaffected. */
  void * engine;
} pll_partition_t;

/* one pruning step (src/optimize/pll_optimize.c:758-765,
   test/src/optimize/blopt-minimal.c:104-111) */
typedef struct pll_operation
{
  unsigned int parent_clv_index;
  int parent_scaler_index;
  unsigned int child1_clv_index;
  unsigned int child1_matrix_index;
  int child1_scaler_index;
  unsigned int child2_clv_index;
  unsigned int child2_matrix_index;
  int child2_scaler_index;
} pll_operation_t;

/* unrooted tree node: a tip is one record (next == NULL), an inner node is a
   ring of three records linked by `next`
   (test/src/optimize/blopt-minimal.c:123-139) */
typedef struct pll_unode_s
{
  char * label;
  double length;
  unsigned int node_index;
  unsigned int clv_index;
  int scaler_index;
  unsigned int pmatrix_index;
  struct pll_unode_s * next;
  struct pll_unode_s * back;
  void * data;
} pll_unode_t;

/* (src/tree/treeinfo.c:91-120: tip_count, inner_count, edge_count, binary,
   nodes, vroot) */
typedef struct pll_utree_s
{
  unsigned int tip_count;
  unsigned int inner_count;
  unsigned int edge_count;
  int binary;
  pll_unode_t ** nodes;
  pll_unode_t * vroot;
} pll_utree_t;

/* rooted tree (only referenced by src/tree/rtree_operations.c and
   src/tree/pll_tree.c rooted helpers; out of the hot path) */
typedef struct pll_rnode_s
{
  char * label;
  double length;
  unsigned int node_index;
  unsigned int clv_index;
  int scaler_index;
  unsigned int pmatrix_index;
  struct pll_rnode_s * left;
  struct pll_rnode_s * right;
  struct pll_rnode_s * parent;
  void * data;
} pll_rnode_t;

typedef struct pll_rtree_s
{
  unsigned int tip_count;
  unsigned int inner_count;
  unsigned int edge_count;
  pll_rnode_t ** nodes;
  pll_rnode_t * root;
} pll_rtree_t;

/* multiple sequence alignment container (src/msa/pll_msa.c) */
typedef struct pll_msa_s
{
  int count;
  int length;
  char ** sequence;
  char ** label;
} pll_msa_t;

/* FASTA reader handle (examples/spr-round/spr-round.c) */
typedef struct pll_fasta
{
  FILE * fp;
  char line[2048];
  const unsigned int * chrstatus;
  long no;
  long filesize;
  long lineno;
  long stripped_count;
  long stripped[256];
} pll_fasta_t;

typedef struct pll_phylip_s
{
  FILE * fp;
  char * line;
  size_t line_size;
  size_t line_maxsize;
  char buffer[2048];
  const unsigned int * chrstatus;
  long no;
  long filesize;
  long lineno;
  long stripped_count;
  long stripped[256];
} pll_phylip_t;

/* re-entrant PRNG state (src/tree/pll_tree.c:903-1277) */
typedef struct pll_random_state_s pll_random_state;

/* parsimony container (src/tree/pll_tree.c:1100-1277; B3, not implemented) */
typedef struct pll_parsimony_s
{
  unsigned int tips;
  unsigned int inner_nodes;
  unsigned int sites;
  unsigned int states;
  unsigned int attributes;
  size_t alignment;
  unsigned int ** packedvector;
  unsigned int * node_cost;
  unsigned int packedvector_count;
  unsigned int const_cost;
  int * informative;
  unsigned int informative_count;
  unsigned int score_buffers;
  unsigned int ancestral_buffers;
  double * score_matrix;
  double * sbuffer;
  unsigned int * anc_states;
} pll_parsimony_t;

typedef struct pll_pars_buildop_s
{
  unsigned int parent_score_index;
  unsigned int child1_score_index;
  unsigned int child2_score_index;
} pll_pars_buildop_t;

typedef struct pll_pars_recop_s
{
  unsigned int node_score_index;
  unsigned int node_ancestral_index;
  unsigned int parent_score_index;
  unsigned int parent_ancestral_index;
} pll_pars_recop_t;

/* ------------------------------------------------------------------ */
/* globals: error reporting (src/pllmod_common.c:42-50)               */
/* ------------------------------------------------------------------ */

PLL_EXPORT extern __thread int pll_errno;
PLL_EXPORT extern __thread char pll_errmsg[200];

/* character maps; only pll_map_nt / pll_map_bin / pll_map_aa are provided */
PLL_EXPORT extern const pll_state_t pll_map_bin[256];
PLL_EXPORT extern const pll_state_t pll_map_nt[256];
PLL_EXPORT extern const pll_state_t pll_map_aa[256];
/* validity classes of the FASTA reader (test/src/tree/treemove-spr.c:178, test/src/binary/binary-random.c:82) */
PLL_EXPORT extern const unsigned int pll_map_fasta[256];

/* ------------------------------------------------------------------ */
/* B1: lifecycle and setters                                          */
/* ------------------------------------------------------------------ */

/* 9-argument constructor (examples/spr-round/spr-round.c:142-150,
   test/src/optimize/blopt-minimal.c:36-44) */
PLL_EXPORT pll_partition_t * pll_partition_create(unsigned int tips,
                                                  unsigned int clv_buffers,
                                                  unsigned int states,
                                                  unsigned int sites,
                                                  unsigned int rate_matrices,
                                                  unsigned int prob_matrices,
                                                  unsigned int rate_cats,
                                                  unsigned int scale_buffers,
                                                  unsigned int attributes);

PLL_EXPORT void pll_partition_destroy(pll_partition_t * partition);

/* char sequence -> state masks through a 256-entry map
   (test/src/optimize/blopt-5states.c:78-80, src/tree/pll_tree.c:1020-1024) */
PLL_EXPORT int pll_set_tip_states(pll_partition_t * partition,
                                  unsigned int tip_index,
                                  const pll_state_t * map,
                                  const char * sequence);

/* reads sites*states (or sites*states_padded if `padding`) doubles and
   replicates each site's vector over all rate categories
   (test/src/optimize/blopt-minimal.c:88-90) */
PLL_EXPORT int pll_set_tip_clv(pll_partition_t * partition,
                               unsigned int tip_index,
                               const double * clv,
                               int padding);

PLL_EXPORT void pll_set_pattern_weights(pll_partition_t * partition,
                                        const unsigned int * pattern_weights);

PLL_EXPORT int pll_set_asc_bias_type(pll_partition_t * partition,
                                     int asc_bias_type);

PLL_EXPORT void pll_set_asc_state_weights(pll_partition_t * partition,
                                          const unsigned int * state_weights);

/* (src/optimize/pll_optimize.c:141, 178, 191, 220, 229, 264) */
PLL_EXPORT void pll_set_subst_params(pll_partition_t * partition,
                                     unsigned int params_index,
                                     const double * params);

PLL_EXPORT void pll_set_frequencies(pll_partition_t * partition,
                                    unsigned int params_index,
                                    const double * frequencies);

PLL_EXPORT void pll_set_category_rates(pll_partition_t * partition,
                                       const double * rates);

PLL_EXPORT void pll_set_category_weights(pll_partition_t * partition,
                                         const double * rate_weights);

PLL_EXPORT int pll_update_eigen(pll_partition_t * partition,
                                unsigned int params_index);

PLL_EXPORT unsigned int pll_count_invariant_sites(pll_partition_t * partition,
                                                  unsigned int * state_inv_count);

/* (src/algorithm/pllmod_algorithm.c, src/msa/pll_msa.c) */
PLL_EXPORT int pll_update_invariant_sites(pll_partition_t * partition);

PLL_EXPORT int pll_update_invariant_sites_proportion(pll_partition_t * partition,
                                                     unsigned int params_index,
                                                     double prop_invar);

/* Yang-1994 discrete gamma (src/optimize/pll_optimize.c:215,
   src/algorithm/algo_callback.c:138) */
PLL_EXPORT int pll_compute_gamma_cats(double alpha,
                                      unsigned int categories,
                                      double * output_rates,
                                      int rates_mode);

/* posix_memalign-compatible: callers release sumtables with plain free()
   (src/tree/treeinfo.c:339, 761-765; src/optimize/pll_optimize.c:1925-1932) */
PLL_EXPORT void * pll_aligned_alloc(size_t size, size_t alignment);
PLL_EXPORT void pll_aligned_free(void * ptr);

PLL_EXPORT unsigned int pll_get_sites_number(const pll_partition_t * partition,
                                             unsigned int clv_index);
PLL_EXPORT unsigned int pll_get_clv_size(const pll_partition_t * partition,
                                         unsigned int clv_index);

/* ------------------------------------------------------------------ */
/* B0: the hot path                                                   */
/* ------------------------------------------------------------------ */

/* replaces libpll-2 pll_update_prob_matrices; call sites
   src/tree/treeinfo.c:854, src/algorithm/algo_search.c:481,
   src/optimize/pll_optimize.c:283, 292, 829, 861, 1316,
   src/tree/pll_tree.c:1957 */
PLL_EXPORT int pll_update_prob_matrices(pll_partition_t * partition,
                                        const unsigned int * params_indices,
                                        const unsigned int * matrix_indices,
                                        const double * branch_lengths,
                                        unsigned int count);

/* replaces libpll-2 pll_update_partials; call sites src/tree/treeinfo.c:1037,
   src/optimize/pll_optimize.c:298, 773, src/tree/pll_tree.c:1980 */
PLL_EXPORT void pll_update_partials(pll_partition_t * partition,
                                    const pll_operation_t * operations,
                                    unsigned int count);

/* call site src/optimize/pll_optimize.c:329 */
PLL_EXPORT double pll_compute_root_loglikelihood(pll_partition_t * partition,
                                                 unsigned int clv_index,
                                                 int scaler_index,
                                                 const unsigned int * freqs_indices,
                                                 double * persite_lnl);

/* call sites src/tree/treeinfo.c:1049, src/optimize/pll_optimize.c:339, 837,
   992, 1057, 1207, src/tree/pll_tree.c:1491 */
PLL_EXPORT double pll_compute_edge_loglikelihood(pll_partition_t * partition,
                                                 unsigned int parent_clv_index,
                                                 int parent_scaler_index,
                                                 unsigned int child_clv_index,
                                                 int child_scaler_index,
                                                 unsigned int matrix_index,
                                                 const unsigned int * freqs_indices,
                                                 double * persite_lnl);

/* call sites src/optimize/pll_optimize.c:800, 1468.  `sumtable` is a
   caller-allocated host buffer that no host code ever reads; in the HIP build
   the pointer is used as the key of a device-resident table. */
PLL_EXPORT int pll_update_sumtable(pll_partition_t * partition,
                                   unsigned int parent_clv_index,
                                   unsigned int child_clv_index,
                                   int parent_scaler_index,
                                   int child_scaler_index,
                                   const unsigned int * params_indices,
                                   double * sumtable);

/* call sites src/optimize/pll_optimize.c:307, 1151, 1249.  Returns the first
   and second derivative of MINUS the log-likelihood at `branch_length`
   (consumer: src/optimize/opt_algorithms.c:208-226). */
PLL_EXPORT int pll_compute_likelihood_derivatives(pll_partition_t * partition,
                                                  int parent_scaler_index,
                                                  int child_scaler_index,
                                                  double branch_length,
                                                  const unsigned int * params_indices,
                                                  const double * sumtable,
                                                  double * d_f,
                                                  double * dd_f);

/* call site src/tree/treeinfo.c:1698 (marginal ancestral states) */
PLL_EXPORT int pll_compute_node_ancestral(pll_partition_t * partition,
                                          unsigned int node_clv_index,
                                          int node_scaler_index,
                                          unsigned int other_clv_index,
                                          int other_scaler_index,
                                          unsigned int matrix_index,
                                          const unsigned int * freqs_indices,
                                          double * ancestral);

/* debugging output (test/src/optimize/blopt-minimal.c:96) */
PLL_EXPORT void pll_show_pmatrix(const pll_partition_t * partition,
                                 unsigned int index,
                                 unsigned int float_precision);

PLL_EXPORT void pll_show_clv(const pll_partition_t * partition,
                             unsigned int clv_index,
                             int scaler_index,
                             unsigned int float_precision);

/* ------------------------------------------------------------------ */
/* B2: unrooted tree utilities                                        */
/* ------------------------------------------------------------------ */

/* (src/tree/treeinfo.c:973-1015) */
PLL_EXPORT int pll_utree_traverse(pll_unode_t * root,
                                  int traversal,
                                  int (*cbtrav)(pll_unode_t *),
                                  pll_unode_t ** outbuffer,
                                  unsigned int * trav_size);

/* NULL is legal for branches / pmatrix_indices / matrix_count
   (src/tree/treeinfo.c:1009-1015) */
PLL_EXPORT void pll_utree_create_operations(pll_unode_t * const* trav_buffer,
                                            unsigned int trav_buffer_size,
                                            double * branches,
                                            unsigned int * pmatrix_indices,
                                            pll_operation_t * ops,
                                            unsigned int * matrix_count,
                                            unsigned int * ops_count);

PLL_EXPORT pll_utree_t * pll_utree_wraptree(pll_unode_t * root,
                                            unsigned int tip_count);

PLL_EXPORT pll_utree_t * pll_utree_wraptree_multi(pll_unode_t * root,
                                                  unsigned int tip_count,
                                                  unsigned int inner_count);

PLL_EXPORT void pll_utree_destroy(pll_utree_t * tree,
                                  void (*cb_destroy)(void *));

PLL_EXPORT pll_unode_t * pll_utree_graph_clone(const pll_unode_t * root);

PLL_EXPORT void pll_utree_graph_destroy(pll_unode_t * root,
                                        void (*cb_destroy)(void *));

PLL_EXPORT pll_utree_t * pll_utree_clone(const pll_utree_t * root);

PLL_EXPORT void pll_utree_reset_template_indices(pll_unode_t * node,
                                                 unsigned int tip_count);

PLL_EXPORT int pll_utree_check_integrity(const pll_utree_t * root);

PLL_EXPORT int pll_utree_every(pll_utree_t * tree,
                               int (*cb)(pll_unode_t *));

PLL_EXPORT pll_utree_t * pll_utree_parse_newick(const char * filename);
PLL_EXPORT pll_utree_t * pll_utree_parse_newick_unroot(const char * filename);
PLL_EXPORT pll_utree_t * pll_utree_parse_newick_string(const char * s);
PLL_EXPORT pll_utree_t * pll_utree_parse_newick_string_unroot(const char * s);

PLL_EXPORT char * pll_utree_export_newick(const pll_unode_t * root,
                                          char * (*cb_serialize)(const pll_unode_t *));

PLL_EXPORT void pll_utree_show_ascii(const pll_unode_t * tree, int options);

/* topology primitives used by src/tree/pll_tree.c:72-293 */
typedef struct pll_utree_rb_s
{
  int move_type;
  union
  {
    struct
    {
      pll_unode_t * p;
      pll_unode_t * r;
      pll_unode_t * rb;
      pll_unode_t * pnb;
      pll_unode_t * pnnb;
      double r_len;
      double pnb_len;
      double pnnb_len;
    } SPR;
    struct
    {
      pll_unode_t * p;
      int nni_type;
    } NNI;
  };
} pll_utree_rb_t;

PLL_EXPORT int pll_utree_spr(pll_unode_t * p, pll_unode_t * r,
                             pll_utree_rb_t * rb,
                             double * branch_lengths,
                             unsigned int * matrix_indices);
PLL_EXPORT int pll_utree_spr_safe(pll_unode_t * p, pll_unode_t * r,
                                  pll_utree_rb_t * rb,
                                  double * branch_lengths,
                                  unsigned int * matrix_indices);
PLL_EXPORT int pll_utree_nni(pll_unode_t * p, int type, pll_utree_rb_t * rb);
PLL_EXPORT int pll_utree_rollback(pll_utree_rb_t * rollback,
                                  double * branch_lengths,
                                  unsigned int * matrix_indices);

/* ------------------------------------------------------------------ */
/* B3: declared so that the remaining pll-modules files parse; the ones
   without an implementation in this engine return PLL_FAILURE / NULL and
   set pll_errno = PLL_ERROR_NOT_IMPLEMENTED.                          */
/* ------------------------------------------------------------------ */

PLL_EXPORT pll_rtree_t * pll_rtree_parse_newick(const char * filename);
PLL_EXPORT void pll_rtree_destroy(pll_rtree_t * tree, void (*cb_destroy)(void *));
PLL_EXPORT char * pll_rtree_export_newick(const pll_rnode_t * root,
                                          char * (*cb_serialize)(const pll_rnode_t *));
PLL_EXPORT void pll_rtree_show_ascii(const pll_rnode_t * tree, int options);
PLL_EXPORT int pll_rtree_traverse(pll_rnode_t * root, int traversal,
                                  int (*cbtrav)(pll_rnode_t *),
                                  pll_rnode_t ** outbuffer,
                                  unsigned int * trav_size);
PLL_EXPORT void pll_rtree_create_operations(pll_rnode_t * const* trav_buffer,
                                            unsigned int trav_buffer_size,
                                            double * branches,
                                            unsigned int * pmatrix_indices,
                                            pll_operation_t * ops,
                                            unsigned int * matrix_count,
                                            unsigned int * ops_count);
PLL_EXPORT pll_rtree_t * pll_rtree_wraptree(pll_rnode_t * root,
                                            unsigned int tip_count);

PLL_EXPORT pll_random_state * pll_random_create(unsigned int seed);
PLL_EXPORT int pll_random_getint(pll_random_state * rstate, int maxval);
PLL_EXPORT void pll_random_destroy(pll_random_state * rstate);

PLL_EXPORT pll_parsimony_t * pll_fastparsimony_init(const pll_partition_t * partition);
PLL_EXPORT void pll_parsimony_destroy(pll_parsimony_t * pars);
PLL_EXPORT pll_utree_t * pll_fastparsimony_stepwise(pll_parsimony_t ** list,
                                                    char * const * labels,
                                                    unsigned int * score,
                                                    unsigned int count,
                                                    unsigned int seed);
PLL_EXPORT int pll_fastparsimony_stepwise_extend(pll_utree_t * tree,
                                                 pll_parsimony_t ** list,
                                                 unsigned int count,
                                                 char * const * labels,
                                                 unsigned int * tip_msa_idmap,
                                                 unsigned int seed,
                                                 unsigned int * score);
PLL_EXPORT int pll_fastparsimony_stepwise_spr_round(pll_utree_t * tree,
                                                    pll_parsimony_t ** pars_list,
                                                    unsigned int pars_count,
                                                    const unsigned int * tip_msa_idmap,
                                                    unsigned int seed,
                                                    const int * clv_valid,
                                                    unsigned int * cost);

PLL_EXPORT pll_fasta_t * pll_fasta_open(const char * filename,
                                        const unsigned int * map);
PLL_EXPORT int pll_fasta_getnext(pll_fasta_t * fd, char ** head,
                                 long * head_len, char ** seq,
                                 long * seq_len, long * seqno);
PLL_EXPORT void pll_fasta_close(pll_fasta_t * fd);
PLL_EXPORT int pll_fasta_rewind(pll_fasta_t * fd);
PLL_EXPORT pll_msa_t * pll_phylip_load(const char * fname, pll_bool_t interleaved);
PLL_EXPORT void pll_msa_destroy(pll_msa_t * msa);
PLL_EXPORT unsigned int * pll_compress_site_patterns(char ** sequence,
                                                     const pll_state_t * map,
                                                     int count,
                                                     int * length);

#ifdef __cplusplus
}
#endif

#endif /* PLLHIP_PLL_H_INCLUDED */
