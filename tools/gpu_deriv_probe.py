#!/usr/bin/env python3
"""Times the derivative scan alone (one sumtable, many scans) for a family and site count:
   python tools/gpu_deriv_probe.py <states> <sites> [K ...]
Run it under rocprofv3 --kernel-trace --stats for kernel times; prints host-side us per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import numpy as np  # noqa: E402
import pllhip_ctypes as pc  # noqa: E402


def main():
    S, N = int(sys.argv[1]), int(sys.argv[2])
    Ks = [int(x) for x in sys.argv[3:]] or [1, 3]
    lib = pc.PllLib(pc.PRODUCT_LIB)
    inst = pc.build_instance(lib, states=S, rate_cats=4, ntips=8, nsites=N, coded=True)
    with inst:
        pc.full_traversal(inst)
        t = inst.tree
        st = inst.alloc_sumtable()
        a = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        inst.update_sumtable(*a, st)
        for K in Ks:
            ts = list(np.geomspace(0.01, 1.0, K))
            inst.derivatives_multi(a[2], a[3], ts, st)
            t0 = time.perf_counter()
            for _ in range(200):
                inst.derivatives_multi(a[2], a[3], ts, st)
            print(f"S={S} N={N} K={K}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (host)")
        t0 = time.perf_counter()
        for _ in range(200):
            inst.edge_lnl(t.root_a, a[2], t.root_b, a[3], t.root_matrix)
        print(f"S={S} N={N} edge lnL: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (host)")
        inst.free_sumtable(st)


if __name__ == "__main__":
    main()
