"""Worker of tests/test_multirank.py: one rank of a gloo process group on CPU.

Each rank owns a contiguous slice of the alignment sites (all nodes' CLVs for
that slice, a replica of the model) and calls the library exactly as a
single-rank caller would; the only exchange is the reference's reduce callback
(signature src/tree/pll_tree.h:274-276, ops 0/1/2 = SUM/MAX/MIN) after every
scalar-returning call -- here implemented with torch.distributed (gloo) so that
the N>1 control flow is exercised without a GPU.  The library under test is the
CPU oracle (test infrastructure).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import pllhip_ctypes as pc  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = pc.PllLib((os.environ.get("PLLHIP_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so")))
    states, R, ntips, nsites = int(sys.argv[1]), 4, 9, 1001

    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}

    def reduce_cb(ctx, data, n, op):
        t = torch.from_numpy(np.ctypeslib.as_array(data, shape=(n,)))   # in place, like the reference
        dist.all_reduce(t, op=ops[op])
    cb = pc.REDUCE_CB(reduce_cb)

    def reduced(values, op=0):
        buf = np.array(values, dtype=np.float64)
        cb(None, buf.ctypes.data_as(pc.c_double_p), len(buf), op)
        return buf

    tree = pc.Tree(ntips)
    codes = pc.random_codes(ntips, nsites, states)
    cmap = pc.state_charmap(states)
    lo, hi = nsites * rank // world, nsites * (rank + 1) // world

    def build(sl):
        inst = pc.build_instance(lib, states=states, rate_cats=R, ntips=ntips, nsites=sl.stop - sl.start,
                                 coded=True, tree=tree)
        for t in range(ntips):
            inst.set_tip_states(t, cmap, (codes[t, sl] + 48).tobytes())
        w = (pc.splitmix64(9, nsites) % np.uint64(5)).astype(np.uint32)[sl]
        inst.set_pattern_weights(w)
        return inst

    mine = build(slice(lo, hi))
    lnl = reduced([pc.full_traversal(mine)])[0]
    st = mine.alloc_sumtable()
    a = (tree.root_a, tree.root_b, tree.scaler_of(tree.root_a), tree.scaler_of(tree.root_b))
    mine.update_sumtable(*a, st)
    d = reduced(mine.derivatives(a[2], a[3], 0.123, st))
    mx = reduced([float(rank + 1), -float(rank)], op=1)
    mn = reduced([float(rank + 1)], op=2)
    wsum = reduced([float(mine.p.contents.pattern_weight_sum)])[0]
    out = {"rank": rank, "lnl": lnl, "df": d[0], "ddf": d[1], "max": list(mx), "min": list(mn), "wsum": wsum}
    if rank == 0:
        full = build(slice(0, nsites))
        out["full_lnl"] = pc.full_traversal(full)
        stf = full.alloc_sumtable()
        full.update_sumtable(*a, stf)
        out["full_df"], out["full_ddf"] = full.derivatives(a[2], a[3], 0.123, stf)
        out["full_wsum"] = float(full.p.contents.pattern_weight_sum)
    with open(os.path.join(sys.argv[2], f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
