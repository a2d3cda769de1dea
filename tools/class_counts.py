#!/usr/bin/env python3
"""tools/class_counts.py <config> [random|simulated] [sites]: classes of sites under every inner node of a bench
configuration's tree (numpy, on the CPU): what the engine's class numbering will find, and which limit stops it"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import pllhip_ctypes as pc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
data = sys.argv[2] if len(sys.argv) > 2 else "random"
states, rate_cats, ntips, nsites = pc.CONFIGS[cfg]
if len(sys.argv) > 3:
    nsites = int(sys.argv[3])
tree = pc.Tree(ntips, 42, 43)
codes = pc.simulated_codes(tree, nsites, states) if data == "simulated" else pc.random_codes(ntips, nsites, states, 44)
cls = {t: codes[t].astype(np.int64) for t in range(ntips)}
ncls = {t: int(codes[t].max()) + 1 for t in range(ntips)}
rows = []
for op in tree.ops:
    parent, c1, c2 = op[0], op[2], op[5]
    key = cls[c1] * ncls[c2] + cls[c2]
    u, inv = np.unique(key, return_inverse=True)
    cls[parent], ncls[parent] = inv.astype(np.int64), len(u)
    rows.append((parent, c1, c2, ncls[c1], ncls[c2], len(u)))
    del cls[c1], cls[c2]
hist = sorted(r[5] for r in rows)
print(cfg, data, nsites, "sites; classes per inner node (sorted):")
print(hist)
for lim in (65536, nsites // 4, nsites // 2):
    print("nodes with <=", lim, "classes:", sum(1 for h in hist if h <= lim), "of", len(hist))
