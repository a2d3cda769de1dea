"""Worker of test_eval_driver.py::test_device_newton_with_a_capped_reduction_grid: the 4-state device Newton-Raphson
loop keeps the sumtable in registers when a thread makes at most two trips over it WITH THE GRID THAT IS LAUNCHED
(newton_capacity, pll_core.hip); PLLHIP_REDUCE_BLOCKS caps that grid (it is read once, when the library is loaded:
hence a process of its own).  argv: <sites> <PLLHIP_REDUCE_BLOCKS>; prints one JSON line with the iterates of the device
loop and of the host loop (one pll_compute_likelihood_derivatives call per iterate) from three starting lengths."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    nsites, blocks = int(sys.argv[1]), int(sys.argv[2])
    os.environ["PLLHIP_REDUCE_BLOCKS"] = str(blocks)
    import pllhip_ctypes as pc
    from test_eval_driver import _host_newton
    lib = pc.PllLib(pc.PRODUCT_LIB)
    out = {"device": [], "host": []}
    with pc.build_instance(lib, states=4, rate_cats=4, ntips=9, nsites=nsites, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        st = a.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, sa, sb, st)
        for start in (float(t.brlens[t.root_matrix]), 1e-4, 5.0):
            try:
                x, trail = _host_newton(lambda v: a.derivatives(sa, sb, v, st), start, 1e-4, 10.0, 1e-5, 32)
                out["host"].append([x] + trail)
            except OverflowError:
                out["host"].append(None)
            try:
                x, its, trail = a.newton_branch(sa, sb, st, start, 1e-4, 10.0, 1e-5, 32)
                out["device"].append([x] + list(trail))
            except RuntimeError:
                out["device"].append(lib.errno)
        a.free_sumtable(st)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
