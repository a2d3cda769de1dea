/*
 * pll_random.c -- re-entrant PRNG handle used by pll-modules' random tree
 * builder (src/tree/pll_tree.c:725-760, 1989-2008).  splitmix64; the stream is
 * NOT the one libpll-2 produces (that source is unavailable), so random trees
 * differ from the reference's for the same seed.
 */
#include "pll.h"

struct pll_random_state_s { unsigned long long s; };

pll_random_state * pll_random_create(unsigned int seed)
{
  pll_random_state * r = (pll_random_state *)malloc(sizeof(*r));
  if (!r)
  {
    pll_errno = PLL_ERROR_MEM_ALLOC;
    snprintf(pll_errmsg, 200, "Cannot allocate PRNG state");
    return NULL;
  }
  r->s = 0x9E3779B97F4A7C15ULL * (seed + 1ULL);
  return r;
}

int pll_random_getint(pll_random_state * r, int maxval)
{
  unsigned long long z = (r->s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (maxval > 0) ? (int)(z % (unsigned long long)maxval) : 0;
}

void pll_random_destroy(pll_random_state * r) { free(r); }
