"""Worker of test_gpu_results.py::test_device_newton_that_cannot_get_its_workgroups: the device-resident
Newton-Raphson loop (include/pllhip.h, pllhip_newton_branch) waits inside its launch for ALL its workgroups; one
that never arrives (PLLHIP_FAULT=newton_stall: workgroup 0 leaves at once, which is what a device shared with other
work does to a grid sized for an empty chip) must end the launch with PLLHIP_ERROR_NEWTON_STUCK after the bounded
wait, leave the engine usable, and the driver must redo the branch on the host loop.
argv: <states> <stall: 0|1>; prints one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))


def main():
    states, stall = int(sys.argv[1]), int(sys.argv[2])
    if stall:
        os.environ["PLLHIP_FAULT"] = "newton_stall"
        os.environ["PLLHIP_FAULT_COUNT"] = "2"              # the direct call below and the driver's first branch
        os.environ["PLLHIP_NEWTON_SPIN_LIMIT"] = "20000"
    import pllhip_ctypes as pc
    lib = pc.PllLib(pc.PRODUCT_LIB)
    out = {}
    # (1) the entry point itself
    with pc.build_instance(lib, states=states, rate_cats=4, ntips=10, nsites=5000 if states <= 20 else 800, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        st = a.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, sa, sb, st)
        before = a.derivatives(sa, sb, 0.1, st)
        try:
            out["direct"] = list(a.newton_branch(sa, sb, st, 0.1, 1e-4, 10.0, 1e-5, 30)[:2])
            out["direct_errno"] = 0
        except RuntimeError:
            out["direct_errno"] = lib.errno
        # the engine's reductions still work (tickets were reset), and the loop itself works again
        out["deriv_unchanged"] = a.derivatives(sa, sb, 0.1, st) == before
        out["lnl_after"] = pc.full_traversal(a)
        a.free_sumtable(st)
    # (2) through the driver: the first branch meets the second injected stall, falls back, and stays on the host loop
    tree = pc.Tree(12, 42, 43)
    subst, freqs = (pc.protein_model() if states == 20 else pc.codon_model() if states == 61 else (pc.DNA_GTR_RATES, pc.DNA_FREQS))
    with pc.Evaluation(lib, tree.newick()) as ev:
        nsites = 4000 if states <= 20 else 600
        ev.add_partition(0, states, nsites, 4, pc.simulated_codes(tree, nsites, states, 45), subst, freqs, 0.7)
        out["lnl0"] = ev.loglh()
        out["lnl1"] = ev.optimize_branches(iters=2)
        out["newick"] = ev.newick()
        out["iterations"] = ev.newton_iterations()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
