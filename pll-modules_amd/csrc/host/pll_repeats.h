/*
 * pll_repeats.h -- the `repeats` member of a partition created with PLL_ATTRIB_SITE_REPEATS.
 * Host code shared by the HIP library and the CPU oracle (internal; include/pll.h declares the type).
 */
#ifndef PLL_REPEATS_H_INCLUDED
#define PLL_REPEATS_H_INCLUDED

#include "pll.h"

#ifdef __cplusplus
extern "C" {
#endif

int pll_repeats_attach(pll_partition_t * partition);     /* PLL_SUCCESS / PLL_FAILURE (out of memory) */
void pll_repeats_release(pll_partition_t * partition);

#ifdef __cplusplus
}
#endif

#endif
