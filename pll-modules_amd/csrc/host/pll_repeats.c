/*
 * pll_repeats.c -- what pll-modules finds behind `partition->repeats`.
 *
 * With PLL_ATTRIB_SITE_REPEATS set in partition->attributes the reference dereferences
 * partition->repeats without a NULL test: the checkpoint code writes and reads
 * pernode_ids / perscale_ids / pernode_allocated_clvs and, for a node that has repeats
 * (pernode_ids[i] != 0), its two index arrays (src/binary/binary_io_operations.c:231-236,
 * 265-282, 329-390; src/binary/pll_binary.c:388-406, 523-550, 579-626, 655, 747, 841-851), and the
 * empirical-frequency code looks a tip's sites up through pernode_site_id when pernode_ids[tip] != 0
 * (src/msa/pll_msa.c:108-112).
 *
 * This engine keeps its classes of sites on the device (csrc/kernels_repeats.hpp); the vectors a caller can
 * see -- host mirrors, pllhip_get_clv, a checkpoint -- are always site-indexed.  In the reference's terms:
 * NO node has repeats.  So every partition created with the attribute carries a table that says exactly that:
 * pernode_ids[i] = 0 and perscale_ids[i] = 0 ("not compressed"), pernode_allocated_clvs[i] = the full site
 * count (what pll_get_sites_number returns, so that the dump never asks for a reallocation), all per-node index
 * arrays NULL.  The reference's walkers then take their "no repeats on this node" arms and read full vectors.
 * Every array is plain malloc memory: the loader frees and reallocates them itself (pll_binary.c:388-406).
 */
#include "pll_repeats.h"
#include <stdlib.h>

static unsigned int never_enable(struct pll_partition * partition, unsigned int left_clv, unsigned int right_clv)
{
  (void)partition; (void)left_clv; (void)right_clv;
  return 0;                                    /* "do not compress this node" */
}

/* the vectors of this library are always allocated for every site: only the record changes */
static void keep_allocation(struct pll_partition * partition, unsigned int clv_index, int scaler_index,
                            unsigned int sites_to_alloc)
{
  (void)scaler_index;
  if (partition->repeats && clv_index < partition->nodes)
    partition->repeats->pernode_allocated_clvs[clv_index] = sites_to_alloc;
}

int pll_repeats_attach(pll_partition_t * p)
{
  unsigned int i;
  const unsigned int nodes = p->nodes ? p->nodes : 1, scalers = p->scale_buffers ? p->scale_buffers : 1;
  pll_repeats_t * r = (pll_repeats_t *)calloc(1, sizeof(*r));
  if (!r) return PLL_FAILURE;
  p->repeats = r;
  r->pernode_ids = (unsigned int *)calloc(nodes, sizeof(unsigned int));
  r->perscale_ids = (unsigned int *)calloc(scalers, sizeof(unsigned int));
  r->pernode_allocated_clvs = (unsigned int *)calloc(nodes, sizeof(unsigned int));
  r->pernode_site_id = (unsigned int **)calloc(nodes, sizeof(unsigned int *));
  r->pernode_id_site = (unsigned int **)calloc(nodes, sizeof(unsigned int *));
  r->enable_repeats = never_enable;
  r->reallocate_repeats = keep_allocation;
  if (!r->pernode_ids || !r->perscale_ids || !r->pernode_allocated_clvs || !r->pernode_site_id || !r->pernode_id_site)
  {
    pll_repeats_release(p);
    return PLL_FAILURE;
  }
  for (i = 0; i < p->nodes; ++i) r->pernode_allocated_clvs[i] = pll_get_sites_number(p, i);
  return PLL_SUCCESS;
}

void pll_repeats_release(pll_partition_t * p)
{
  unsigned int i;
  pll_repeats_t * r = p->repeats;
  if (!r) return;
  /* (the per-node arrays are the loader's: pll_binary.c:848-851 mallocs them one by one) */
  for (i = 0; i < p->nodes; ++i)
  {
    if (r->pernode_site_id) free(r->pernode_site_id[i]);
    if (r->pernode_id_site) free(r->pernode_id_site[i]);
  }
  free(r->pernode_site_id);
  free(r->pernode_id_site);
  free(r->pernode_ids);
  free(r->perscale_ids);
  free(r->pernode_allocated_clvs);
  free(r);
  p->repeats = NULL;
}
