"""Worker of tests/test_multirank.py::test_c_driver_*: one rank of a gloo group running the
C evaluation driver (include/pllhip_eval.h: pllhip_eval_set_parallel_context, NULL partition
slots) the way a pll-modules worker runs treeinfo (src/tree/treeinfo.c:215-227, 1024-1067):

  mode "sites"  every rank holds all partitions, each with its own slice of the sites
  mode "parts"  every rank holds only some partitions; the others are NULL slots
  prefix "scaled-" / "unlinked-": branch-length linkage (per-partition scalers / lengths)

The only exchange is the reduce callback (gloo here, so it runs without a GPU or with two
processes on one GPU).  argv: <lib: oracle|product> <mode> <outdir>
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

PARTS = [(4, 4, 700), (20, 4, 333), (4, 2, 501)]     # (states, rate cats, sites)
NTIPS = 9


def model_of(states):
    if states == 4:
        return pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841
    r, f = pc.protein_model()
    return r, f, 0.5


def build(lib, tree, owned, site_range, cb=None, flags=0, linkage=0):
    """Evaluation over PARTS; partition k is present if k in owned, holding sites
    site_range(k) = (lo, hi) of its alignment"""
    ev = pc.Evaluation(lib, tree.newick(), flags=flags, nparts=len(PARTS))
    for k, (states, R, nsites) in enumerate(PARTS):
        if k not in owned:
            ev.add_remote_partition(k)
            continue
        lo, hi = site_range(k, nsites)
        subst, freqs, alpha = model_of(states)
        codes = pc.simulated_codes(tree, nsites, states, seed=45 + k)[:, lo:hi]
        ev.add_partition(k, states, hi - lo, R, codes, subst, freqs, alpha, coded=True)
    if cb is not None:
        ev.set_parallel_context(cb)
    if linkage:
        ev.set_linkage(linkage, [0.7, 1.0, 1.9])
    return ev


def run(ev):
    out = {"lnl": ev.loglh()}
    out["lnl_opt"] = ev.optimize_branches(bl_min=1e-4, bl_max=10.0, eps=0.01, iters=4)
    out["lnl_after"] = ev.loglh()
    out["newick"] = ev.newick()
    out["tree_lengths"] = [ev.partition_tree_length(p) for p in range(len(PARTS))]
    out["scans"], out["iterations"] = ev.counters()[2], ev.newton_iterations()
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    which, mode, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
    path = pc.PRODUCT_LIB if which == "product" else (os.environ.get("PLLHIP_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so"))
    lib = pc.PllLib(path)
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}
    calls = []

    def reduce_cb(ctx, data, n, op):
        calls.append((n, op))
        t = torch.from_numpy(np.ctypeslib.as_array(data, shape=(n,)))   # in place, like the reference
        dist.all_reduce(t, op=ops[op])
    cb = pc.REDUCE_CB(reduce_cb)

    tree = pc.Tree(NTIPS, 42, 43)
    linkage = 2 if mode.startswith("unlinked") else 1 if mode.startswith("scaled") else 0
    mode = mode.split("-")[-1]
    if mode == "sites":
        owned = set(range(len(PARTS)))
        rng = lambda k, n: (n * rank // world, n * (rank + 1) // world)
    else:
        owned = {k for k in range(len(PARTS)) if k % world == rank}
        rng = lambda k, n: (0, n)
    # "sites": several trial lengths per scan everywhere (PLLHIP_EVAL_ALWAYS_SPECULATE), so the
    # {df, ddf} messages carry more than one length; "parts": the policy the libraries ask for
    flags = 4 if mode == "sites" else 0
    with build(lib, tree, owned, rng, cb, flags, linkage) as ev:
        out = run(ev)
    out["reduce_calls"] = len(calls)
    out["payloads"] = sorted(set(n for n, _ in calls))
    if rank == 0:
        with build(lib, tree, set(range(len(PARTS))), lambda k, n: (0, n), None, flags, linkage) as ev:
            out["single"] = run(ev)
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
