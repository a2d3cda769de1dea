"""pllhip_newton_branch_multi at the C4 slice sizes: which forms (register-resident / streaming instances) complete.
usage: gpu_newton_multi_probe.py [sizes ...]   env: PLLHIP_NEWTON_MAX_SHARE, PLLHIP_NEWTON_RESIDENT, PLLHIP_NEWTON_SPIN_LIMIT"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
os.environ.setdefault("PLLHIP_NEWTON_SPIN_LIMIT", "300000")
import pllhip_ctypes as pc

lib = pc.PllLib(pc.PRODUCT_LIB)
spec = [(4, 31250), (4, 31250), (20, 15625), (20, 15625)]
if len(sys.argv) > 1:
    spec = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
NT = int(os.environ.get("PROBE_TAXA", "12"))
tree = pc.Tree(NT, 42, 43)
insts = [pc.build_instance(lib, states=s_, rate_cats=4, ntips=NT, nsites=n_, coded=True, tree=tree, seed_shift=k) for k, (s_, n_) in enumerate(spec)]
sts = []
for a in insts:
    a.tree = tree
    pc.full_traversal(a)
    sts.append(a.alloc_sumtable())
sa, sb = tree.scaler_of(tree.root_a), tree.scaler_of(tree.root_b)
for a, st in zip(insts, sts):
    a.update_sumtable(tree.root_a, tree.root_b, sa, sb, st)
for rep in range(3):
    t0 = time.perf_counter()
    try:
        x, its, trail = pc.newton_branch_multi(lib, insts, sa, sb, sts, None, 0.1, 1e-4, 10.0, 1e-5, 32)
        print(f"ok   x {x!r} iterations {its} {1e6 * (time.perf_counter() - t0) / max(1, its):.1f} us per iterate", flush=True)
    except RuntimeError as exc:
        print(f"FAIL {exc} after {time.perf_counter() - t0:.3f} s", flush=True)
for k, a in enumerate(insts):
    try:
        x, its, trail = a.newton_branch(sa, sb, sts[k], 0.1, 1e-4, 10.0, 1e-5, 32)
        print(f"single partition {k}: x {x!r} iterations {its}", flush=True)
    except RuntimeError as exc:
        print(f"single partition {k}: FAIL {exc}", flush=True)
