// kernels_s20.hpp -- 20-state (protein) kernel family.
#pragma once

#include "kernels_common.hpp"
#include "kernels_generic.hpp"
#include "engine.h"

namespace pllhip {

static int launch_partials_s20(Engine * e, const OpBatch & batch, unsigned nops)
{
  return launch_partials_generic(e, batch, nops);
}

static int launch_edge_lnl_s20(Engine * e, const ModelView & mv, const ParamIdx & fidx,
                               const NodeRef & parent, const NodeRef & child,
                               const double * pm, const double * lut,
                               const unsigned * ps, const unsigned * cs,
                               double * persite, unsigned nblocks)
{
  return launch_edge_lnl_generic(e, mv, fidx, parent, child, pm, lut, ps, cs, persite, nblocks);
}

static int launch_sumtable_s20(Engine * e, const ModelView & mv, const ParamIdx & params,
                               const NodeRef & parent, const NodeRef & child, double * d_sum)
{
  return launch_sumtable_generic(e, mv, params, parent, child, d_sum);
}

static int launch_derivatives_s20(Engine * e, const ModelView & mv, const ParamIdx & params, double t,
                                  const double * d_sum, const unsigned * ps, const unsigned * cs,
                                  unsigned nblocks)
{
  return launch_derivatives_generic(e, mv, params, t, d_sum, ps, cs, nblocks);
}

} // namespace pllhip
