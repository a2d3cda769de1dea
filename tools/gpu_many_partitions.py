#!/usr/bin/env python3
"""Many small partitions under one tree (a phylogenomic data set: one partition per gene): time per
evaluation for P partitions of `sites` sites each.  tools/gpu_many_partitions.py <states> <P> <sites> [taxa]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import bench
import pllhip_ctypes as pc

states, P, sites = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
taxa = int(sys.argv[4]) if len(sys.argv) > 4 else 100
per_branch = os.environ.get("PMATRIX_CALLS", "per-branch") == "per-branch"   # batched: one call per partition
lib = pc.PllLib(pc.PRODUCT_LIB)
tree = pc.Tree(taxa, 42, 43)
ev, insts = bench.make_evaluation(pc, lib, tree, [(states, sites)] * P, 4, 44, per_branch)
for _ in range(3): l = ev.loglh()
t0 = time.perf_counter()
steps = 10
for _ in range(steps): l = ev.loglh()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"states": states, "partitions": P, "sites_each": sites, "taxa": taxa, "ms_per_evaluation": dt * 1e3,
                  "site_updates_per_s": (taxa - 2) * P * sites * 4 / dt, "lnl": l,
                  "pmatrix_calls": "per-branch" if per_branch else "batched"}))
ev.close()
