#!/usr/bin/env python3
"""Host-side cost of the entry points (time until the call returns; the kernels run asynchronously) on a small
partition, where they dominate: tools/gpu_host_overheads.py [states] [sites]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import numpy as np
import pllhip_ctypes as pc
states = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
lib = pc.PllLib(pc.PRODUCT_LIB)
out = {"states": states, "sites": sites}
with pc.build_instance(lib, states=states, rate_cats=4, ntips=100, nsites=sites, coded=True) as a:
    t = a.tree
    pc.full_traversal(a)
    ops = a.make_ops(t.ops_with_scalers(True)); nops = len(t.ops)
    mi = pc._u32(np.arange(t.nedges)); bl = pc._f64(t.brlens)
    ptrs = [(pc.C.cast(mi.ctypes.data + 4 * k, pc.c_uint_p), pc.C.cast(bl.ctypes.data + 8 * k, pc.c_double_p)) for k in range(t.nedges)]
    fn, p, pp = a.L.pll_update_prob_matrices, a.p, a.params_p
    reps = 200
    lib.lib.pllhip_synchronize(a.p)
    t0 = time.perf_counter()
    for _ in range(reps):
        for m, b in ptrs: fn(p, pp, m, b, 1)
    out["us_per_pmatrix_call_incl_python"] = (time.perf_counter() - t0) / (reps * t.nedges) * 1e6
    lib.lib.pllhip_synchronize(a.p)
    t0 = time.perf_counter()
    for _ in range(reps): fn(p, pp, mi.ctypes.data_as(pc.c_uint_p), bl.ctypes.data_as(pc.c_double_p), t.nedges)
    out["us_per_batched_pmatrix_call"] = (time.perf_counter() - t0) / reps * 1e6
    lib.lib.pllhip_synchronize(a.p)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(p, pp, mi.ctypes.data_as(pc.c_uint_p), bl.ctypes.data_as(pc.c_double_p), t.nedges)
        a.update_partials(ops, nops)
    out["us_per_batched_pmatrix_plus_full_update_partials_call"] = (time.perf_counter() - t0) / reps * 1e6
    lib.lib.pllhip_synchronize(a.p)
    t0 = time.perf_counter()
    for _ in range(reps):
        pc.full_traversal(a)
    out["us_per_full_evaluation_incl_wait"] = (time.perf_counter() - t0) / reps * 1e6
print(json.dumps(out))
