#!/usr/bin/env python3
"""Per-kernel microbenchmarks on the GPU box: single operations of each child
type (tip x tip, tip x inner, inner x inner), edge lnL, sumtable, derivatives,
timed with the engine's HIP-event hook / wall clock after a sync; prints
algorithmic GB/s (SURVEY.md section 8d byte counts)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import numpy as np  # noqa: E402
import pllhip_ctypes as pc  # noqa: E402

NONE = pc.PLL_SCALE_BUFFER_NONE


def main():
    states = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nsites = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lib = pc.PllLib(pc.PRODUCT_LIB)
    R = 4
    inst = pc.build_instance(lib, states=states, rate_cats=R, ntips=8, nsites=nsites, coded=True)
    L = lib.lib
    t = inst.tree
    pc.full_traversal(inst)
    S, N = states, nsites
    inner = [op for op in t.ops]
    tt = next(op for op in inner if op[2] < 8 and op[5] < 8)
    ti = next(op for op in inner if (op[2] < 8) != (op[5] < 8))
    ii = next((op for op in inner if op[2] >= 8 and op[5] >= 8), None)
    cases = [("tip x tip", tt, N * R * 8 * S + 2 * N + 4 * N),
             ("tip x inner", ti, N * R * 16 * S + N + 8 * N)]
    if ii:
        cases.append(("inner x inner", ii, N * R * 24 * S + 12 * N))
    prof = pc.Profile()
    for name, op, nbytes in cases:
        for scal in (True, False):
            o = op if scal else (op[0], NONE, op[2], op[3], NONE, op[5], op[6], NONE)
            arr = inst.make_ops([o])
            inst.update_partials(arr, 1)
            L.pllhip_profile_partials(inst.p, 1)
            for _ in range(reps):
                inst.update_partials(arr, 1)
            L.pllhip_profile_read(inst.p, C.byref(prof))
            L.pllhip_profile_partials(inst.p, 0)
            ms = prof.kernel_ms / reps
            print(f"S={S} {name:14s} scalers={int(scal)}  {ms * 1e3:9.1f} us/op  "
                  f"{nbytes / ms / 1e6:8.1f} GB/s alg  {N * R / ms / 1e6:8.2f} G site-upd/s")
    # several independent ops in one launch (level batching)
    sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)

    def timed(fn, n=reps):
        fn()
        L.pllhip_synchronize(inst.p)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        L.pllhip_synchronize(inst.p)
        return (time.perf_counter() - t0) / n
    dt = timed(lambda: inst.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix))
    nb = N * R * 8 * S * (2 if t.root_b >= 8 else 1)
    print(f"S={S} edge lnL (sync each call)      {dt * 1e6:9.1f} us  {nb / dt / 1e9:8.1f} GB/s alg")
    if ii:    # an inner-inner branch (timing only: the parent CLV looks the other way)
        dt = timed(lambda: inst.edge_lnl(ii[0], t.scaler_of(ii[0]), ii[2], t.scaler_of(ii[2]), ii[3]))
        print(f"S={S} edge lnL inner-inner (sync)      {dt * 1e6:9.1f} us  {N * R * 16 * S / dt / 1e9:8.1f} GB/s alg")
    st = inst.alloc_sumtable()
    dt = timed(lambda: inst.update_sumtable(t.root_a, t.root_b, sa, sb, st))
    print(f"S={S} sumtable                       {dt * 1e6:9.1f} us  {(nb + N * R * 8 * S) / dt / 1e9:8.1f} GB/s alg")
    dt = timed(lambda: inst.derivatives(sa, sb, 0.1, st))
    print(f"S={S} derivatives (sync each call)   {dt * 1e6:9.1f} us  {N * R * 8 * S / dt / 1e9:8.1f} GB/s alg")
    dt = timed(lambda: inst.update_pmatrices(np.arange(t.nedges), t.brlens, one_by_one=True), 5)
    print(f"S={S} {t.nedges} P-matrix calls, one per branch  {dt * 1e6:9.1f} us total, {dt / t.nedges * 1e6:6.2f} us each")
    dt = timed(lambda: inst.update_pmatrices(np.arange(t.nedges), t.brlens), 5)
    print(f"S={S} {t.nedges} P-matrices in one call          {dt * 1e6:9.1f} us")
    inst.free_sumtable(st)
    inst.close()


if __name__ == "__main__":
    main()
