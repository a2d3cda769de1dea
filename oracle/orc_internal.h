/*
 * oracle/orc_internal.h -- shared declarations of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a plain-C, fp64, single-threaded
 * restatement of the likelihood hot path behind include/pll.h.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (pll-modules_amd/) never links or calls anything in this directory.
 *
 * Parity status: PINNED for S=4 and S=5 without scaling by the two
 * self-contained golden files of the reference test-suite
 * (test/out/optimize/blopt-minimal.out, blopt-5states.out; see
 * tests/test_oracle_golden.py).  UNPINNED by the reference for: per-site
 * scaling, p-inv, 20/61 states, >3 taxa (every such reference test needs
 * alignments that its Makefile downloads, test/Makefile:59-72).  The arithmetic
 * itself lives in xflouris/libpll-2 (git submodule, branch master, no pinned
 * commit, absent from /root/reference), so the formulas follow SURVEY.md
 * section 8a / Appendix B (textbook Felsenstein 1981, Yang 1994).
 */
#ifndef ORC_INTERNAL_H_
#define ORC_INTERNAL_H_

#include "pll.h"

void orc_set_error(int code, const char * fmt, ...);

/* symmetric eigen-decomposition by cyclic Jacobi rotations.
   a: n*n row-major symmetric (destroyed), w: eigenvalues, v: n*n row-major,
   column k of v is eigenvector k. */
void orc_jacobi_eigen(long double * a, unsigned int n, long double * w, long double * v);

/* (re)build eigenvecs/inv_eigenvecs/eigenvals of one rate matrix */
int orc_update_eigen(pll_partition_t * p, unsigned int params_index);

/* tip value helper: CLV entry (site n, rate r, state j) of any node */
/* sites the arrays hold: the alignment patterns plus, with ascertainment-bias correction, one
   constant pattern per state (site sites + k: every tip shows state k) */
static inline unsigned int orc_salloc(const pll_partition_t * p)
{
  return p->sites + (p->asc_bias_alloc ? (unsigned int)p->asc_additional_sites : 0u);
}

static inline double orc_clv_at(const pll_partition_t * p, unsigned int node,
                                unsigned int n, unsigned int r, unsigned int j)
{
  if (node < p->tips && (p->attributes & PLL_ATTRIB_PATTERN_TIP))
  {
    pll_state_t m = (p->states == 4) ? (pll_state_t)p->tipchars[node][n]
                                     : p->tipmap[p->tipchars[node][n]];
    return (double)((m >> j) & 1ULL);
  }
  return p->clv[node][((size_t)n * p->rate_cats + r) * p->states_padded + j];
}

#endif
