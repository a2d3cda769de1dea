/*
 * oracle/orc_partition.c -- partition lifecycle, setters and tip loaders of
 * the CPU oracle (TEST INFRASTRUCTURE ONLY, see orc_internal.h).
 *
 * Follows the call contracts visible in the reference:
 *   constructor arguments          test/src/optimize/blopt-minimal.c:36-44
 *   setters                        src/optimize/pll_optimize.c:141-264
 *   tip loaders                    test/src/optimize/blopt-minimal.c:88-90,
 *                                  test/src/optimize/blopt-5states.c:78-80
 *   CLV layout [site][rate][state] src/msa/pll_msa.c:114-124
 *   tipchars / tipmap convention   src/msa/pll_msa.c:66-103
 *   error convention               src/pllmod_common.c:42-50
 */
#include "orc_internal.h"
#include "../pll-modules_amd/csrc/host/pll_repeats.h"
#include <stdarg.h>

__thread int pll_errno = 0;
__thread char pll_errmsg[200] = {0};

void orc_set_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, sizeof(pll_errmsg), fmt, ap);
  va_end(ap);
}

void * pll_aligned_alloc(size_t size, size_t alignment)
{
  void * mem = NULL;
  if (alignment < sizeof(void *)) alignment = sizeof(void *);
  if (posix_memalign(&mem, alignment, size ? size : alignment)) return NULL;
  return mem;
}

void pll_aligned_free(void * ptr) { free(ptr); }

/* libpll-2: the sites a node's vector is allocated for (its classes when it has repeats; no node of this
   library has: ../pll-modules_amd/csrc/host/pll_repeats.c), plus the ascertainment-bias columns */
unsigned int pll_get_sites_number(const pll_partition_t * p, unsigned int clv_index)
{
  unsigned int n = (p->repeats && clv_index < p->nodes) ? p->repeats->pernode_ids[clv_index] : 0u;
  return orc_salloc(p) - p->sites + (n ? n : p->sites);
}

unsigned int pll_get_clv_size(const pll_partition_t * p, unsigned int clv_index)
{
  (void)clv_index;
  return orc_salloc(p) * p->rate_cats * p->states_padded;
}

static double ** alloc_rows(unsigned int rows, size_t cols)
{
  unsigned int i;
  double ** t = (double **)calloc(rows ? rows : 1, sizeof(double *));
  if (!t) return NULL;
  for (i = 0; i < rows; ++i)
    if (!(t[i] = (double *)calloc(cols ? cols : 1, sizeof(double)))) return NULL;
  return t;
}

pll_partition_t * pll_partition_create(unsigned int tips,
                                       unsigned int clv_buffers,
                                       unsigned int states,
                                       unsigned int sites,
                                       unsigned int rate_matrices,
                                       unsigned int prob_matrices,
                                       unsigned int rate_cats,
                                       unsigned int scale_buffers,
                                       unsigned int attributes)
{
  unsigned int i;
  pll_partition_t * p;

  if (states < 2 || states > 64 || !rate_cats || !rate_matrices)
  {
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid partition dimensions");
    return NULL;
  }

  p = (pll_partition_t *)calloc(1, sizeof(*p));
  if (!p) goto nomem;
  p->tips = tips;
  p->clv_buffers = clv_buffers;
  p->nodes = tips + clv_buffers;
  p->states = states;
  p->sites = sites;
  p->pattern_weight_sum = sites;
  p->rate_matrices = rate_matrices;
  p->prob_matrices = prob_matrices;
  p->rate_cats = rate_cats;
  p->scale_buffers = scale_buffers;
  p->attributes = attributes;
  p->alignment = PLL_ALIGNMENT_CPU;
  p->states_padded = states;
  /* ascertainment-bias correction: one extra constant pattern per state behind the alignment
     (libpll-2: asc_bias_alloc / asc_additional_sites; src/tree/treeinfo.c:333-337 and
     src/binary/binary_io_operations.c size their buffers for it) */
  if (attributes & (PLL_ATTRIB_AB_FLAG | PLL_ATTRIB_AB_MASK))
  {
    p->asc_bias_alloc = 1;
    p->asc_additional_sites = (int)states;
  }
  const unsigned int salloc = sites + (p->asc_bias_alloc ? states : 0);

  size_t clv_len = (size_t)salloc * rate_cats * p->states_padded;
  p->clv = (double **)calloc(p->nodes ? p->nodes : 1, sizeof(double *));
  if (!p->clv) goto nomem;
  for (i = 0; i < p->nodes; ++i)
  {
    if (i < tips && (attributes & PLL_ATTRIB_PATTERN_TIP)) continue;
    if (!(p->clv[i] = (double *)calloc(clv_len ? clv_len : 1, sizeof(double))))
      goto nomem;
  }

  size_t pm_len = (size_t)rate_cats * states * p->states_padded;
  p->pmatrix = (double **)calloc(prob_matrices ? prob_matrices : 1, sizeof(double *));
  if (!p->pmatrix) goto nomem;
  if (prob_matrices)
  {
    double * block = (double *)calloc(pm_len * prob_matrices, sizeof(double));
    if (!block) goto nomem;
    for (i = 0; i < prob_matrices; ++i) p->pmatrix[i] = block + pm_len * i;
  }

  p->rates = (double *)calloc(rate_cats, sizeof(double));
  p->rate_weights = (double *)calloc(rate_cats, sizeof(double));
  p->prop_invar = (double *)calloc(rate_matrices, sizeof(double));
  p->eigen_decomp_valid = (int *)calloc(rate_matrices, sizeof(int));
  p->pattern_weights = (unsigned int *)calloc(salloc ? salloc : 1, sizeof(unsigned int));
  if (!p->rates || !p->rate_weights || !p->prop_invar ||
      !p->eigen_decomp_valid || !p->pattern_weights) goto nomem;
  for (i = 0; i < rate_cats; ++i)
  {
    p->rates[i] = 1.0;
    p->rate_weights[i] = 1.0 / rate_cats;
  }
  for (i = 0; i < salloc; ++i) p->pattern_weights[i] = 1;

  p->subst_params = alloc_rows(rate_matrices, (size_t)states * (states - 1) / 2);
  p->frequencies = alloc_rows(rate_matrices, p->states_padded);
  p->eigenvecs = alloc_rows(rate_matrices, (size_t)states * p->states_padded);
  p->inv_eigenvecs = alloc_rows(rate_matrices, (size_t)states * p->states_padded);
  p->eigenvals = alloc_rows(rate_matrices, p->states_padded);
  if (!p->subst_params || !p->frequencies || !p->eigenvecs ||
      !p->inv_eigenvecs || !p->eigenvals) goto nomem;

  p->scale_buffer = (unsigned int **)calloc(scale_buffers ? scale_buffers : 1,
                                            sizeof(unsigned int *));
  if (!p->scale_buffer) goto nomem;
  for (i = 0; i < scale_buffers; ++i)
    if (!(p->scale_buffer[i] = (unsigned int *)calloc((salloc ? salloc : 1) *
                                                      ((attributes & PLL_ATTRIB_RATE_SCALERS) ? (size_t)rate_cats : 1),
                                                      sizeof(unsigned int))))
      goto nomem;

  if (attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    p->tipchars = (unsigned char **)calloc(tips ? tips : 1, sizeof(unsigned char *));
    p->charmap = (unsigned char *)calloc(PLL_ASCII_SIZE, 1);
    p->tipmap = (pll_state_t *)calloc(PLL_ASCII_SIZE, sizeof(pll_state_t));
    if (!p->tipchars || !p->charmap || !p->tipmap) goto nomem;
    for (i = 0; i < tips; ++i)
      if (!(p->tipchars[i] = (unsigned char *)calloc(salloc ? salloc : 1, 1)))
        goto nomem;
    if (states == 4)
    {
      /* DNA: the code is the 4-bit mask itself (src/msa/pll_msa.c:66-82) */
      for (i = 0; i < 16; ++i) p->tipmap[i] = i;
      p->maxstates = 16;
    }
  }
  /* the reference dereferences partition->repeats whenever the attribute is set */
  if ((attributes & PLL_ATTRIB_SITE_REPEATS) && !pll_repeats_attach(p)) goto nomem;
  return p;

nomem:
  orc_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
  if (p) pll_partition_destroy(p);
  return NULL;
}

static void free_rows(void ** t, unsigned int rows)
{
  unsigned int i;
  if (!t) return;
  for (i = 0; i < rows; ++i) free(t[i]);
  free(t);
}

void pll_partition_destroy(pll_partition_t * p)
{
  if (!p) return;
  free_rows((void **)p->clv, p->nodes);
  if (p->pmatrix) { free(p->pmatrix[0]); free(p->pmatrix); }
  free(p->rates);
  free(p->rate_weights);
  free(p->prop_invar);
  free(p->eigen_decomp_valid);
  free(p->pattern_weights);
  free(p->invariant);
  free_rows((void **)p->subst_params, p->rate_matrices);
  free_rows((void **)p->frequencies, p->rate_matrices);
  free_rows((void **)p->eigenvecs, p->rate_matrices);
  free_rows((void **)p->inv_eigenvecs, p->rate_matrices);
  free_rows((void **)p->eigenvals, p->rate_matrices);
  free_rows((void **)p->scale_buffer, p->scale_buffers);
  free_rows((void **)p->tipchars, p->tips);
  free(p->charmap);
  free(p->tipmap);
  pll_aligned_free(p->ttlookup);
  pll_repeats_release(p);
  free(p);
}

/* libpll keeps a tip-tip lookup table of (2^ceil(log2 maxstates))^2 * states_padded *
   rate_cats doubles with coded tips; this engine does not use one, but the reference's
   binary dump writes it (src/binary/binary_io_operations.c:242-250): keep a zero-filled
   array of that size */
static int ensure_ttlookup(pll_partition_t * p)
{
  unsigned int l2 = 0;
  size_t n;
  while ((1u << l2) < p->maxstates) ++l2;
  n = ((size_t)1 << (2 * l2)) * p->states_padded * p->rate_cats;
  pll_aligned_free(p->ttlookup);
  p->ttlookup = (double *)pll_aligned_alloc(n * sizeof(double), p->alignment);
  if (!p->ttlookup)
  {
    orc_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate the tip-tip lookup placeholder");
    return PLL_FAILURE;
  }
  memset(p->ttlookup, 0, n * sizeof(double));
  return PLL_SUCCESS;
}

void pll_set_subst_params(pll_partition_t * p, unsigned int idx, const double * v)
{
  memcpy(p->subst_params[idx], v,
         sizeof(double) * p->states * (p->states - 1) / 2);
  p->eigen_decomp_valid[idx] = 0;
}

void pll_set_frequencies(pll_partition_t * p, unsigned int idx, const double * v)
{
  memcpy(p->frequencies[idx], v, sizeof(double) * p->states);
  p->eigen_decomp_valid[idx] = 0;
}

void pll_set_category_rates(pll_partition_t * p, const double * v)
{
  memcpy(p->rates, v, sizeof(double) * p->rate_cats);
}

void pll_set_category_weights(pll_partition_t * p, const double * v)
{
  memcpy(p->rate_weights, v, sizeof(double) * p->rate_cats);
}

void pll_set_pattern_weights(pll_partition_t * p, const unsigned int * w)
{
  unsigned int i;
  memcpy(p->pattern_weights, w, sizeof(unsigned int) * p->sites);
  p->pattern_weight_sum = 0;
  for (i = 0; i < p->sites; ++i) p->pattern_weight_sum += w[i];
}

/* Lewis: lnL - W log(1 - sum_k L_k); Felsenstein: lnL + w log(sum_k L_k), w = sum of the state
   weights; Stamatakis: lnL + sum_k w_k log(L_k)  (Leache et al. 2015; [libpll-2 knowledge]) */
int pll_set_asc_bias_type(pll_partition_t * p, int t)
{
  if (!p->asc_bias_alloc)
  {
    orc_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Partition was not created for ascertainment bias correction");
    return PLL_FAILURE;
  }
  if (t != PLL_ATTRIB_AB_LEWIS && t != PLL_ATTRIB_AB_FELSENSTEIN && t != PLL_ATTRIB_AB_STAMATAKIS)
  {
    orc_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm");
    return PLL_FAILURE;
  }
  p->attributes = (p->attributes & ~(unsigned int)PLL_ATTRIB_AB_MASK) | (unsigned int)t;
  return PLL_SUCCESS;
}

void pll_set_asc_state_weights(pll_partition_t * p, const unsigned int * w)
{
  if (!p->asc_bias_alloc) return;
  memcpy(p->pattern_weights + p->sites, w, sizeof(unsigned int) * p->states);
}

int pll_set_tip_states(pll_partition_t * p, unsigned int tip,
                       const pll_state_t * map, const char * seq)
{
  unsigned int n, j, r;
  const unsigned int old_codes = p->maxstates;
  if (tip >= p->tips)
  {
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid tip index %u", tip);
    return PLL_FAILURE;
  }
  for (n = 0; n < p->sites; ++n)
  {
    unsigned char c = (unsigned char)seq[n];
    pll_state_t m = map[c];
    if (!m || (p->states < 64 && (m >> p->states)))
    {
      orc_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE,
                    "Illegal state code in tip \"%c\"", seq[n]);
      return PLL_FAILURE;
    }
    if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
    {
      unsigned int code;
      if (p->states == 4)
        code = (unsigned int)m;
      else
      {
        for (code = 0; code < p->maxstates; ++code)
          if (p->tipmap[code] == m) break;
        if (code == p->maxstates)
        {
          if (code >= PLL_ASCII_SIZE)
          {
            orc_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Too many tip codes");
            return PLL_FAILURE;
          }
          p->tipmap[code] = m;
          p->maxstates++;
        }
      }
      p->charmap[c] = (unsigned char)code;
      p->tipchars[tip][n] = (unsigned char)code;
    }
    else
    {
      double * v = p->clv[tip] + (size_t)n * p->rate_cats * p->states_padded;
      for (r = 0; r < p->rate_cats; ++r)
        for (j = 0; j < p->states_padded; ++j)
          v[r * p->states_padded + j] =
              (j < p->states) ? (double)((m >> j) & 1ULL) : 0.0;
    }
  }
  /* ascertainment-bias columns: pattern sites + k shows state k at every tip */
  for (n = p->sites; n < orc_salloc(p); ++n)
  {
    const pll_state_t m = 1ULL << (n - p->sites);
    if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
    {
      unsigned int code;
      if (p->states == 4)
        code = (unsigned int)m;
      else
      {
        for (code = 0; code < p->maxstates; ++code)
          if (p->tipmap[code] == m) break;
        if (code == p->maxstates)
        {
          if (code >= PLL_ASCII_SIZE)
          {
            orc_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Too many tip codes");
            return PLL_FAILURE;
          }
          p->tipmap[code] = m;
          p->maxstates++;
        }
      }
      p->tipchars[tip][n] = (unsigned char)code;
    }
    else
    {
      double * v = p->clv[tip] + (size_t)n * p->rate_cats * p->states_padded;
      for (r = 0; r < p->rate_cats; ++r)
        for (j = 0; j < p->states_padded; ++j)
          v[r * p->states_padded + j] = (j == n - p->sites) ? 1.0 : 0.0;
    }
  }
  if ((p->attributes & PLL_ATTRIB_PATTERN_TIP) && (!p->ttlookup || p->maxstates != old_codes))
    return ensure_ttlookup(p);
  return PLL_SUCCESS;
}

int pll_set_tip_clv(pll_partition_t * p, unsigned int tip, const double * clv,
                    int padding)
{
  unsigned int n, r;
  if (tip >= p->tips)
  {
    orc_set_error(PLL_ERROR_PARAM_INVALID, "Invalid tip index %u", tip);
    return PLL_FAILURE;
  }
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    orc_set_error(PLL_ERROR_TIPDATA_ILLEGALFUNCTION,
                  "Cannot use pll_set_tip_clv with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }
  /* one S-vector per site in, replicated over the rate categories */
  unsigned int in_stride = padding ? p->states_padded : p->states;
  for (n = 0; n < p->sites; ++n)
    for (r = 0; r < p->rate_cats; ++r)
    {
      double * dst = p->clv[tip] + ((size_t)n * p->rate_cats + r) * p->states_padded;
      memset(dst, 0, sizeof(double) * p->states_padded);
      memcpy(dst, clv + (size_t)n * in_stride, sizeof(double) * p->states);
    }
  for (n = p->sites; n < orc_salloc(p); ++n)        /* ascertainment-bias columns */
    for (r = 0; r < p->rate_cats; ++r)
    {
      double * dst = p->clv[tip] + ((size_t)n * p->rate_cats + r) * p->states_padded;
      memset(dst, 0, sizeof(double) * p->states_padded);
      dst[n - p->sites] = 1.0;
    }
  return PLL_SUCCESS;
}

/* --- invariant sites ------------------------------------------------ */

/* a site is invariant for state s if every tip is compatible with s; record
   the lowest such state, or -1 */
int pll_update_invariant_sites(pll_partition_t * p)
{
  unsigned int n, t, j;
  if (!p->invariant)
    p->invariant = (int *)malloc(sizeof(int) * (p->sites ? p->sites : 1));
  if (!p->invariant)
  {
    orc_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate invariant sites array");
    return PLL_FAILURE;
  }
  for (n = 0; n < p->sites; ++n)
  {
    pll_state_t common = ~0ULL;
    for (t = 0; t < p->tips; ++t)
    {
      pll_state_t m = 0;
      for (j = 0; j < p->states; ++j)
        if (orc_clv_at(p, t, n, 0, j) > 0.0) m |= (1ULL << j);
      common &= m;
    }
    if (p->states < 64) common &= ((1ULL << p->states) - 1);
    p->invariant[n] = common ? (int)PLL_STATE_CTZ(common) : -1;
  }
  return PLL_SUCCESS;
}

unsigned int pll_count_invariant_sites(pll_partition_t * p, unsigned int * state_inv_count)
{
  unsigned int n, count = 0;
  int had = (p->invariant != NULL);
  if (state_inv_count) memset(state_inv_count, 0, sizeof(unsigned int) * p->states);
  if (!had && !pll_update_invariant_sites(p)) return 0;
  for (n = 0; n < p->sites; ++n)
    if (p->invariant[n] >= 0)
    {
      count += p->pattern_weights[n];
      if (state_inv_count) state_inv_count[p->invariant[n]] += p->pattern_weights[n];
    }
  if (!had) { free(p->invariant); p->invariant = NULL; }
  return count;
}

int pll_update_invariant_sites_proportion(pll_partition_t * p, unsigned int idx,
                                          double prop_invar)
{
  if (idx >= p->rate_matrices)
  {
    orc_set_error(PLL_ERROR_INVAR_PARAMINDEX, "Invalid params index");
    return PLL_FAILURE;
  }
  if (prop_invar < 0 || prop_invar >= 1)
  {
    orc_set_error(PLL_ERROR_INVAR_PROPORTION, "Invalid proportion of invariant sites");
    return PLL_FAILURE;
  }
  if (prop_invar > 0 && !p->invariant)
    if (!pll_update_invariant_sites(p)) return PLL_FAILURE;
  p->prop_invar[idx] = prop_invar;
  return PLL_SUCCESS;
}

/* --- debugging output (format of test/out/optimize/blopt-minimal.out) -- */

void pll_show_pmatrix(const pll_partition_t * p, unsigned int index,
                      unsigned int prec)
{
  unsigned int r, i, j;
  for (r = 0; r < p->rate_cats; ++r)
  {
    const double * m = p->pmatrix[index] + (size_t)r * p->states * p->states_padded;
    for (i = 0; i < p->states; ++i)
    {
      for (j = 0; j < p->states; ++j)
        printf("%+2.*f   ", prec, m[i * p->states_padded + j]);
      printf("\n");
    }
    printf("\n");
  }
}

void pll_show_clv(const pll_partition_t * p, unsigned int clv_index,
                  int scaler_index, unsigned int prec)
{
  unsigned int n, r, j;
  (void)scaler_index;
  printf("[ ");
  for (n = 0; n < p->sites; ++n)
  {
    printf("{");
    for (r = 0; r < p->rate_cats; ++r)
    {
      printf("(");
      for (j = 0; j < p->states; ++j)
        printf("%.*f%s", prec, orc_clv_at(p, clv_index, n, r, j),
               j + 1 < p->states ? "," : "");
      printf(")%s", r + 1 < p->rate_cats ? "," : "");
    }
    printf("} ");
  }
  printf("]\n");
}
