"""Worker of the failure-path tests: what happens to the OTHER ranks when one rank fails.

argv: <lib: oracle|product> <mode> <outdir>
  mode "eval"      two gloo ranks run the C driver with the reduce callback; rank 1's second evaluation fails
                   locally (PLLHIP_EVAL_FAULT).  Every rank must see NaN in that very call -- the failing rank
                   still takes part in the reduction, with NaN -- and nobody hangs; the third evaluation works.
  modes "deposit" / "collective" / "publish"   (product only, one process, RCCL world of one) the library's own
                   communicator with an injected failure (PLLHIP_FAULT): see include/pllhip.h, pllhip_results_fetch
Every rank writes rank<k>.json and exits with code 3 when it saw the failure (a worker of a real run exits).
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    which, mode, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
    rank = int(os.environ.get("RANK", "0"))
    if mode == "eval":
        if rank == 1:
            os.environ["PLLHIP_EVAL_FAULT"] = "2"
    else:
        os.environ["PLLHIP_FAULT"] = {"deposit": "deposit@5", "collective": "collective@2", "publish": "publish@2"}[mode]
        os.environ["PLLHIP_COLLECTIVE_TIMEOUT_S"] = "1"
    import pllhip_ctypes as pc
    import _evaldriver_worker as W
    path = pc.PRODUCT_LIB if which == "product" else (os.environ.get("PLLHIP_ORACLE_LIB") or
                                                      os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so"))
    lib = pc.PllLib(path)
    tree = pc.Tree(W.NTIPS, 42, 43)
    out = {"rank": rank, "lnl": [], "errno": [], "errmsg": []}

    def evaluate(ev):
        lib.errno = 0
        v = ev.L.pllhip_eval_loglh(ev.ev, 0)
        out["lnl"].append(None if math.isnan(v) else v)
        out["errno"].append(lib.errno)
        out["errmsg"].append(lib.errmsg if lib.errno else "")

    if mode == "eval":
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        world = dist.get_world_size()
        ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}

        def reduce_cb(ctx, data, n, op):
            t = torch.from_numpy(np.ctypeslib.as_array(data, shape=(n,)))
            dist.all_reduce(t, op=ops[op])
        cb = pc.REDUCE_CB(reduce_cb)
        rng = lambda k, n: (n * rank // world, n * (rank + 1) // world)
        with W.build(lib, tree, set(range(len(W.PARTS))), rng, cb) as ev:
            for _ in range(3):
                evaluate(ev)
        with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
            json.dump(out, f)
        dist.barrier()
        dist.destroy_process_group()
    else:
        import ctypes as C
        idbuf = C.create_string_buffer(128)
        assert lib.lib.pllhip_comm_get_unique_id(idbuf)
        comm = lib.lib.pllhip_comm_create(idbuf.raw, 0, 1, 0)
        assert comm, lib.errmsg
        with W.build(lib, tree, set(range(len(W.PARTS))), lambda k, n: (0, n)) as ev:
            ev.attach_comm(comm)
            for _ in range(4):
                evaluate(ev)
        with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
            json.dump(out, f)
        lib.lib.pllhip_comm_destroy(comm)
    sys.exit(3 if any(v is None for v in out["lnl"]) else 0)


if __name__ == "__main__":
    main()
