// pll_partition.cpp -- host side of the partition behind include/pll.h, tier B1:
// lifecycle, setters, tip loaders, invariant-site bookkeeping, error globals.
//
// Contracts follow the reference call sites:
//   pll_partition_create (9 args)      examples/spr-round/spr-round.c:142-150
//   setters                            src/optimize/pll_optimize.c:141-264
//   pll_set_tip_states / _clv          test/src/optimize/blopt-5states.c:78-80,
//                                      test/src/optimize/blopt-minimal.c:88-90
//   tipchars / tipmap convention       src/msa/pll_msa.c:66-103
//   pll_aligned_alloc + free()         src/tree/treeinfo.c:339, 761-765
//   pll_errno / pll_errmsg             src/pllmod_common.c:42-50
//
// Big arrays (CLVs, scalers) are never allocated on the host here: the pointer
// tables partition->clv / scale_buffer exist (pll-modules indexes them) but
// hold NULL until pllhip_sync_to_host() materialises them.
#include "engine.h"
#include "host/pll_repeats.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <new>
#include <unordered_map>
#include <vector>

extern "C" {
__thread int pll_errno = 0;
__thread char pll_errmsg[200] = {0};
}

namespace pllhip {

void set_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, sizeof(pll_errmsg), fmt, ap);
  va_end(ap);
}

static unsigned padded_states(unsigned s)
{
  return (s <= 2) ? s : ((s + 3u) & ~3u);
}

template <typename T>
static T ** alloc_table(unsigned rows, size_t cols)
{
  T ** t = static_cast<T **>(calloc(rows ? rows : 1, sizeof(T *)));
  if (!t) return nullptr;
  for (unsigned i = 0; i < rows; ++i)
    if (!(t[i] = static_cast<T *>(calloc(cols ? cols : 1, sizeof(T))))) return nullptr;
  return t;
}

template <typename T>
static void free_table(T ** t, unsigned rows)
{
  if (!t) return;
  for (unsigned i = 0; i < rows; ++i) free(t[i]);
  free(t);
}

} // namespace pllhip

using namespace pllhip;

extern "C" {

void * pll_aligned_alloc(size_t size, size_t alignment)
{
  void * mem = nullptr;
  if (alignment < sizeof(void *)) alignment = sizeof(void *);
  if (posix_memalign(&mem, alignment, size ? size : alignment)) return nullptr;
  return mem;
}

void pll_aligned_free(void * ptr) { free(ptr); }

// libpll-2: the sites a node's vector is allocated for -- its classes when it has repeats, else every site (+ the
// ascertainment-bias columns).  No node of this library has repeats in the reference's sense (host/pll_repeats.c).
unsigned int pll_get_sites_number(const pll_partition_t * p, unsigned int clv_index)
{
  unsigned int n = (p->repeats && clv_index < p->nodes) ? p->repeats->pernode_ids[clv_index] : 0u;
  if (!n) n = p->sites;
  return n + (p->asc_bias_alloc ? (unsigned)p->asc_additional_sites : 0u);
}

unsigned int pll_get_clv_size(const pll_partition_t * p, unsigned int)
{
  return (p->sites + (p->asc_bias_alloc ? p->asc_additional_sites : 0)) * p->rate_cats * p->states_padded;
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
  if (states < 2 || states > 64 || !rate_cats || rate_cats > MAX_RATE_CATS || !rate_matrices)
  {
    set_error(PLL_ERROR_PARAM_INVALID,
              "Invalid partition dimensions (2 <= states <= 64, 1 <= rate_cats <= %u)",
              MAX_RATE_CATS);
    return nullptr;
  }

  pll_partition_t * p = static_cast<pll_partition_t *>(calloc(1, sizeof(*p)));
  if (!p)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return nullptr;
  }
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
  p->alignment = PLL_ALIGNMENT_HIP;
  p->states_padded = padded_states(states);
  const unsigned Sp = p->states_padded;
  // ascertainment-bias correction: one constant pattern per state behind the alignment in every
  // per-site array (libpll-2's asc_bias_alloc / asc_additional_sites; pll-modules sizes its
  // sumtables and dumps for it: src/tree/treeinfo.c:333-337, src/binary/binary_io_operations.c)
  if (attributes & (PLL_ATTRIB_AB_FLAG | PLL_ATTRIB_AB_MASK))
  {
    p->asc_bias_alloc = 1;
    p->asc_additional_sites = static_cast<int>(states);
  }
  const unsigned salloc = sites + (p->asc_bias_alloc ? states : 0u);

  bool ok = true;
  p->clv = static_cast<double **>(calloc(p->nodes ? p->nodes : 1, sizeof(double *)));
  p->scale_buffer = static_cast<unsigned int **>(calloc(scale_buffers ? scale_buffers : 1,
                                                        sizeof(unsigned int *)));
  const size_t pm_len = static_cast<size_t>(rate_cats) * states * Sp;
  p->pmatrix = static_cast<double **>(calloc(prob_matrices ? prob_matrices : 1, sizeof(double *)));
  ok = ok && p->clv && p->scale_buffer && p->pmatrix;
  if (ok && prob_matrices)
  {
    double * block = static_cast<double *>(calloc(pm_len * prob_matrices, sizeof(double)));
    ok = block != nullptr;
    for (unsigned i = 0; ok && i < prob_matrices; ++i) p->pmatrix[i] = block + pm_len * i;
  }
  p->rates = static_cast<double *>(calloc(rate_cats, sizeof(double)));
  p->rate_weights = static_cast<double *>(calloc(rate_cats, sizeof(double)));
  p->prop_invar = static_cast<double *>(calloc(rate_matrices, sizeof(double)));
  p->eigen_decomp_valid = static_cast<int *>(calloc(rate_matrices, sizeof(int)));
  p->pattern_weights = static_cast<unsigned int *>(calloc(salloc ? salloc : 1, sizeof(unsigned int)));
  ok = ok && p->rates && p->rate_weights && p->prop_invar && p->eigen_decomp_valid &&
       p->pattern_weights;
  if (ok)
  {
    for (unsigned i = 0; i < rate_cats; ++i)
    {
      p->rates[i] = 1.0;
      p->rate_weights[i] = 1.0 / rate_cats;
    }
    for (unsigned i = 0; i < salloc; ++i) p->pattern_weights[i] = 1;
  }
  p->subst_params = alloc_table<double>(rate_matrices, static_cast<size_t>(states) * (states - 1) / 2);
  p->frequencies = alloc_table<double>(rate_matrices, Sp);
  p->eigenvecs = alloc_table<double>(rate_matrices, static_cast<size_t>(states) * Sp);
  p->inv_eigenvecs = alloc_table<double>(rate_matrices, static_cast<size_t>(states) * Sp);
  p->eigenvals = alloc_table<double>(rate_matrices, Sp);
  ok = ok && p->subst_params && p->frequencies && p->eigenvecs && p->inv_eigenvecs && p->eigenvals;

  if (ok && (attributes & PLL_ATTRIB_PATTERN_TIP))
  {
    p->tipchars = alloc_table<unsigned char>(tips, salloc);
    p->charmap = static_cast<unsigned char *>(calloc(PLL_ASCII_SIZE, 1));
    p->tipmap = static_cast<pll_state_t *>(calloc(PLL_ASCII_SIZE, sizeof(pll_state_t)));
    ok = p->tipchars && p->charmap && p->tipmap;
    if (ok && states == 4)
    {
      // DNA: the code is the 4-bit mask itself (src/msa/pll_msa.c:66-82)
      for (unsigned i = 0; i < 16; ++i) p->tipmap[i] = i;
      p->maxstates = 16;
    }
  }
  if (!ok)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    pll_partition_destroy(p);
    return nullptr;
  }

  if ((attributes & PLL_ATTRIB_SITE_REPEATS) && !pll_repeats_attach(p))
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate the site-repeats tables.");
    pll_partition_destroy(p);
    return nullptr;
  }
  p->engine = engine_create(p);   // sets pll_errno on failure
  if (!p->engine)
  {
    pll_partition_destroy(p);
    return nullptr;
  }
  if (attributes & PLLHIP_ATTRIB_HOST_MIRRORS)
  {
    // a loader is going to fill these (include/pllhip.h); calloc'ed pages cost nothing
    // until they are written
    const size_t len = static_cast<size_t>(salloc) * rate_cats * Sp;
    const unsigned first = (attributes & PLL_ATTRIB_PATTERN_TIP) ? tips : 0;
    for (unsigned i = first; ok && i < p->nodes; ++i)
      ok = (p->clv[i] = static_cast<double *>(calloc(len ? len : 1, sizeof(double)))) != nullptr;
    for (unsigned i = 0; ok && i < scale_buffers; ++i)
      ok = (p->scale_buffer[i] = static_cast<unsigned int *>(
                calloc((salloc ? salloc : 1) * ((attributes & PLL_ATTRIB_RATE_SCALERS) ? (size_t)rate_cats : 1),
                       sizeof(unsigned int)))) != nullptr;
    if (!ok)
    {
      set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate the host mirrors.");
      pll_partition_destroy(p);
      return nullptr;
    }
  }
  return p;
}

void pll_partition_destroy(pll_partition_t * p)
{
  if (!p) return;
  if (p->engine) engine_destroy(engine_of(p));
  free_table(p->clv, p->nodes);
  free_table(p->scale_buffer, p->scale_buffers);
  if (p->pmatrix) { free(p->pmatrix[0]); free(p->pmatrix); }
  free(p->rates);
  free(p->rate_weights);
  free(p->prop_invar);
  free(p->eigen_decomp_valid);
  free(p->pattern_weights);
  free(p->invariant);
  free_table(p->subst_params, p->rate_matrices);
  free_table(p->frequencies, p->rate_matrices);
  free_table(p->eigenvecs, p->rate_matrices);
  free_table(p->inv_eigenvecs, p->rate_matrices);
  free_table(p->eigenvals, p->rate_matrices);
  free_table(p->tipchars, p->tips);
  free(p->charmap);
  free(p->tipmap);
  pll_aligned_free(p->ttlookup);
  pll_repeats_release(p);
  free(p);
}

// libpll keeps a tip-tip lookup table of (2^ceil(log2 maxstates))^2 * states_padded *
// rate_cats doubles with coded tips; this engine has per-matrix tables on the device
// instead, but the reference's binary dump writes the array
// (src/binary/binary_io_operations.c:242-250): keep a zero-filled one of that size
static int ensure_ttlookup(pll_partition_t * p)
{
  unsigned l2 = 0;
  while ((1u << l2) < p->maxstates) ++l2;
  const size_t n = (static_cast<size_t>(1) << (2 * l2)) * p->states_padded * p->rate_cats;
  pll_aligned_free(p->ttlookup);
  p->ttlookup = static_cast<double *>(pll_aligned_alloc(n * sizeof(double), p->alignment));
  if (!p->ttlookup)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate the tip-tip lookup placeholder");
    return PLL_FAILURE;
  }
  memset(p->ttlookup, 0, n * sizeof(double));
  return PLL_SUCCESS;
}

void pll_set_subst_params(pll_partition_t * p, unsigned int idx, const double * v)
{
  memcpy(p->subst_params[idx], v, sizeof(double) * p->states * (p->states - 1) / 2);
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
  memcpy(p->pattern_weights, w, sizeof(unsigned int) * p->sites);
  p->pattern_weight_sum = 0;
  for (unsigned i = 0; i < p->sites; ++i) p->pattern_weight_sum += w[i];
  upload_weights(p);
}

// Lewis: lnL - W log(1 - sum_k L_k); Felsenstein: lnL + w log(sum_k L_k), w = sum of the state
// weights; Stamatakis: lnL + sum_k w_k log(L_k)   (Leache et al. 2015; [libpll-2 knowledge])
int pll_set_asc_bias_type(pll_partition_t * p, int type)
{
  if (!p->asc_bias_alloc)
  {
    set_error(PLL_ERROR_AB_INVALIDMETHOD, "Partition was not created for ascertainment bias correction");
    return PLL_FAILURE;
  }
  if (type != PLL_ATTRIB_AB_LEWIS && type != PLL_ATTRIB_AB_FELSENSTEIN && type != PLL_ATTRIB_AB_STAMATAKIS)
  {
    set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm");
    return PLL_FAILURE;
  }
  p->attributes = (p->attributes & ~static_cast<unsigned>(PLL_ATTRIB_AB_MASK)) | static_cast<unsigned>(type);
  return PLL_SUCCESS;
}

void pll_set_asc_state_weights(pll_partition_t * p, const unsigned int * w)
{
  if (!p->asc_bias_alloc) return;
  // host-side only: the correction is applied on the host (the device copy of these weights is 0)
  memcpy(p->pattern_weights + p->sites, w, sizeof(unsigned int) * p->states);
}

int pll_update_eigen(pll_partition_t * p, unsigned int idx)
{
  if (idx >= p->rate_matrices)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Invalid params index");
    return PLL_FAILURE;
  }
  return update_eigen_host(p, idx);
}

static unsigned salloc_of(const pll_partition_t * p)
{
  return p->sites + (p->asc_bias_alloc ? static_cast<unsigned>(p->asc_additional_sites) : 0u);
}

int pll_set_tip_states(pll_partition_t * p, unsigned int tip,
                       const pll_state_t * map, const char * seq)
{
  if (tip >= p->tips)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Invalid tip index %u", tip);
    return PLL_FAILURE;
  }
  const unsigned S = p->states, Sp = p->states_padded, R = p->rate_cats;
  const bool coded = (p->attributes & PLL_ATTRIB_PATTERN_TIP) != 0;
  double * tmp = nullptr;
  if (!coded)
  {
    tmp = static_cast<double *>(malloc(sizeof(double) * (size_t)salloc_of(p) * R * Sp + 8));
    if (!tmp)
    {
      set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate tip staging buffer");
      return PLL_FAILURE;
    }
  }
  unsigned old_codes = p->maxstates;
  std::vector<unsigned> site_class(coded ? 0 : salloc_of(p), 0u);
  std::vector<unsigned long long> masks;
  std::unordered_map<unsigned long long, unsigned> class_of;
  for (unsigned n = 0; n < salloc_of(p); ++n)
  {
    // behind the alignment: the ascertainment-bias column of state n - sites
    const bool asc = n >= p->sites;
    unsigned char c = asc ? 0 : static_cast<unsigned char>(seq[n]);
    pll_state_t m = asc ? (1ULL << (n - p->sites)) : map[c];
    if (!m || (S < 64 && (m >> S)))
    {
      free(tmp);
      set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Illegal state code in tip \"%c\"", seq[n]);
      return PLL_FAILURE;
    }
    if (coded)
    {
      unsigned code;
      if (S == 4)
        code = static_cast<unsigned>(m);
      else
      {
        for (code = 0; code < p->maxstates; ++code)
          if (p->tipmap[code] == m) break;
        if (code == p->maxstates)
        {
          if (code >= PLL_ASCII_SIZE)
          {
            set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Too many distinct tip codes");
            return PLL_FAILURE;
          }
          p->tipmap[code] = m;
          p->maxstates++;
        }
      }
      if (!asc) p->charmap[c] = static_cast<unsigned char>(code);
      p->tipchars[tip][n] = static_cast<unsigned char>(code);
    }
    else
    {
      double * v = tmp + (size_t)n * R * Sp;
      for (unsigned r = 0; r < R; ++r)
        for (unsigned j = 0; j < Sp; ++j)
          v[r * Sp + j] = (j < S) ? static_cast<double>((m >> j) & 1ULL) : 0.0;
      // the class of the site: its state mask (site repeats without pattern tips, upload_tip_classes)
      unsigned k = 0;
      if (class_of.size() <= 64)
      {
        for (; k < masks.size(); ++k) if (masks[k] == m) break;
        if (k == masks.size()) { masks.push_back(m); class_of[m] = k; }
      }
      else
      {
        auto it = class_of.find(m);
        if (it == class_of.end()) { k = (unsigned)masks.size(); masks.push_back(m); class_of[m] = k; }
        else k = it->second;
      }
      site_class[n] = k;
    }
  }
  int rc;
  if (coded)
  {
    if (p->maxstates != old_codes) invalidate_luts(p);
    if ((!p->ttlookup || p->maxstates != old_codes) && !ensure_ttlookup(p)) return PLL_FAILURE;
    rc = upload_tip_codes(p, tip);
  }
  else
  {
    rc = upload_tip_clv(p, tip, tmp);
    free(tmp);
    if (rc) rc = upload_tip_classes(p, tip, site_class.data(), masks.data(), (unsigned)masks.size());
  }
  return rc;
}

int pll_set_tip_clv(pll_partition_t * p, unsigned int tip, const double * clv, int padding)
{
  if (tip >= p->tips)
  {
    set_error(PLL_ERROR_PARAM_INVALID, "Invalid tip index %u", tip);
    return PLL_FAILURE;
  }
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    set_error(PLL_ERROR_TIPDATA_ILLEGALFUNCTION,
              "Cannot use pll_set_tip_clv with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }
  const unsigned S = p->states, Sp = p->states_padded, R = p->rate_cats;
  double * tmp = static_cast<double *>(calloc((size_t)salloc_of(p) * R * Sp + 1, sizeof(double)));
  if (!tmp)
  {
    set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate tip staging buffer");
    return PLL_FAILURE;
  }
  // one S-vector per site in, replicated over the rate categories
  const unsigned in_stride = padding ? Sp : S;
  for (unsigned n = 0; n < p->sites; ++n)
    for (unsigned r = 0; r < R; ++r)
      memcpy(tmp + ((size_t)n * R + r) * Sp, clv + (size_t)n * in_stride, sizeof(double) * S);
  for (unsigned n = p->sites; n < salloc_of(p); ++n)       // ascertainment-bias columns
    for (unsigned r = 0; r < R; ++r) tmp[((size_t)n * R + r) * Sp + (n - p->sites)] = 1.0;
  int rc = upload_tip_clv(p, tip, tmp);
  free(tmp);
  return rc;
}

int pll_update_invariant_sites_proportion(pll_partition_t * p, unsigned int idx, double prop_invar)
{
  if (idx >= p->rate_matrices)
  {
    set_error(PLL_ERROR_INVAR_PARAMINDEX, "Invalid params index");
    return PLL_FAILURE;
  }
  if (prop_invar < 0 || prop_invar >= 1)
  {
    set_error(PLL_ERROR_INVAR_PROPORTION, "Invalid proportion of invariant sites");
    return PLL_FAILURE;
  }
  if (prop_invar > 0 && !p->invariant)
    if (!pll_update_invariant_sites(p)) return PLL_FAILURE;
  p->prop_invar[idx] = prop_invar;
  return PLL_SUCCESS;
}

unsigned int pll_count_invariant_sites(pll_partition_t * p, unsigned int * state_inv_count)
{
  unsigned count = 0;
  const bool had = p->invariant != nullptr;
  if (state_inv_count) memset(state_inv_count, 0, sizeof(unsigned) * p->states);
  if (!had && !pll_update_invariant_sites(p)) return 0;
  for (unsigned n = 0; n < p->sites; ++n)
    if (p->invariant[n] >= 0)
    {
      count += p->pattern_weights[n];
      if (state_inv_count) state_inv_count[p->invariant[n]] += p->pattern_weights[n];
    }
  return count;
}

void pll_show_pmatrix(const pll_partition_t * p, unsigned int index, unsigned int prec)
{
  pllhip_sync_to_host(const_cast<pll_partition_t *>(p), PLLHIP_SYNC_PMATRIX);
  for (unsigned r = 0; r < p->rate_cats; ++r)
  {
    const double * m = p->pmatrix[index] + (size_t)r * p->states * p->states_padded;
    for (unsigned i = 0; i < p->states; ++i)
    {
      for (unsigned j = 0; j < p->states; ++j)
        printf("%+2.*f   ", prec, m[i * p->states_padded + j]);
      printf("\n");
    }
    printf("\n");
  }
}

void pll_show_clv(const pll_partition_t * cp, unsigned int clv_index, int, unsigned int prec)
{
  pll_partition_t * p = const_cast<pll_partition_t *>(cp);
  const size_t len = (size_t)salloc_of(p) * p->rate_cats * p->states_padded;
  double * buf = static_cast<double *>(malloc(sizeof(double) * (len ? len : 1)));
  if (!buf || !pllhip_get_clv(p, clv_index, buf)) { free(buf); return; }
  printf("[ ");
  for (unsigned n = 0; n < p->sites; ++n)
  {
    printf("{");
    for (unsigned r = 0; r < p->rate_cats; ++r)
    {
      printf("(");
      for (unsigned j = 0; j < p->states; ++j)
        printf("%.*f%s", prec, buf[((size_t)n * p->rate_cats + r) * p->states_padded + j],
               j + 1 < p->states ? "," : "");
      printf(")%s", r + 1 < p->rate_cats ? "," : "");
    }
    printf("} ");
  }
  printf("]\n");
  free(buf);
}

} // extern "C"
