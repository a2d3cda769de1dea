#!/usr/bin/env python3
"""Secondary workloads of SURVEY.md section 8d on one GPU, driven through the C ABI:

  W2  Newton-Raphson inner loop: 1 pll_update_sumtable + 16
      pll_compute_likelihood_derivatives on one branch
      (src/optimize/pll_optimize.c:1468, 1249)
  W3  SPR-like scoring: 1000 x {2 pll_update_prob_matrices, a 1-3 op
      pll_update_partials, 1 pll_compute_edge_loglikelihood}
      (src/algorithm/algo_search.c:786-790)
  C4  multi-partition mixed DNA + protein (2 x 250k DNA + 2 x 125k AA sites,
      100 taxa, linked branch lengths): one full evaluation = loop over the
      partitions + sum, as treeinfo_compute_loglh does (src/tree/treeinfo.c:1024-1067)

  SPR one pllhip_eval_spr_round (the counterpart of pllmod_algo_spr_round,
      src/algorithm/algo_search.c:1052-1484) on the codon configuration C5: the
      whole round runs in the C driver, one ctypes call

Every call goes through ctypes, i.e. ~2-5 us of Python per call sit inside the
W3 timings; a C caller pays less.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
import numpy as np  # noqa: E402
import pllhip_ctypes as pc  # noqa: E402

NONE = pc.PLL_SCALE_BUFFER_NONE


def w2(lib, cfg, out):
    S, R, ntips, N = pc.CONFIGS[cfg]
    ntips = min(ntips, 16)
    inst = pc.build_instance(lib, states=S, rate_cats=R, ntips=ntips, nsites=N, coded=True)
    with inst:
        pc.full_traversal(inst)
        t = inst.tree
        # an inner-inner branch near the root
        op = next(o for o in reversed(t.ops) if o[2] >= ntips and o[5] >= ntips)
        p_clv, c_clv = op[0], op[2]
        st = inst.alloc_sumtable()
        args = (p_clv, c_clv, t.scaler_of(p_clv), t.scaler_of(c_clv))
        brl = np.geomspace(1e-3, 2.0, 16)

        def once():
            inst.update_sumtable(*args, st)
            for b in brl:
                inst.derivatives(args[2], args[3], float(b), st)
        once()
        lib.lib.pllhip_synchronize(inst.p)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        dt = (time.perf_counter() - t0) / reps
        nbytes = N * R * 8 * S * (3 + 16)
        out[f"W2_{cfg}"] = {"ms": dt * 1e3, "us_per_derivative_call": (dt * 1e6) / 17,
                            "algorithmic_GBps": nbytes / dt / 1e9,
                            "what": "1 sumtable + 16 derivative evaluations, host sync per derivative"}
        for K in (1, 2, 4, 8):
            tt = [float(b) for b in brl[:K]]
            inst.derivatives_multi(args[2], args[3], tt, st)
            t0 = time.perf_counter()
            for _ in range(50):
                inst.derivatives_multi(args[2], args[3], tt, st)
            out[f"W2_{cfg}"][f"us_per_scan_K{K}"] = (time.perf_counter() - t0) / 50 * 1e6
        inst.free_sumtable(st)


def w3(lib, cfg, out, nsites=None):
    S, R, ntips, N = pc.CONFIGS[cfg]
    N = nsites or N
    inst = pc.build_instance(lib, states=S, rate_cats=R, ntips=ntips, nsites=N, coded=True)
    with inst:
        pc.full_traversal(inst)
        t = inst.tree
        rnd = pc.splitmix64(77, 4000)
        tail = t.ops[-3:]
        sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
        arrs = {k: inst.make_ops(tail[-k:]) for k in (1, 2, 3)}
        mi = np.zeros(2, dtype=np.uint32)
        bl = np.zeros(2)
        mi_p, bl_p = mi.ctypes.data_as(pc.c_uint_p), bl.ctypes.data_as(pc.c_double_p)
        L = lib.lib

        def run(n):
            updates = 0
            for i in range(n):
                k = 1 + int(rnd[3 * i] % np.uint64(3))
                ops = tail[-k:]
                mi[0], mi[1] = ops[0][3], ops[0][6]
                bl[0] = 0.01 + float(rnd[3 * i + 1] % np.uint64(1000)) * 2e-4
                bl[1] = 0.01 + float(rnd[3 * i + 2] % np.uint64(1000)) * 2e-4
                L.pll_update_prob_matrices(inst.p, inst.params_p, mi_p, bl_p, 2)
                inst.update_partials(arrs[k], k)
                inst.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
                updates += k
            return updates
        run(20)
        L.pllhip_synchronize(inst.p)
        t0 = time.perf_counter()
        upd = run(1000)
        dt = time.perf_counter() - t0
        out[f"W3_{cfg}_{N}"] = {"ms_total": dt * 1e3, "us_per_iteration": dt * 1e3,
                                "site_updates_per_s": upd * N * R / dt,
                                "what": "1000 x {2 P-matrix updates, 1-3 partial ops, 1 edge lnL}"}


def c4(lib, out):
    tree = pc.Tree(100)
    parts = [pc.build_instance(lib, states=4, rate_cats=4, ntips=100, nsites=250_000, coded=True, tree=tree, seed_shift=s)
             for s in (0, 1)]
    parts += [pc.build_instance(lib, states=20, rate_cats=4, ntips=100, nsites=125_000, coded=True, tree=tree, seed_shift=s)
              for s in (2, 3)]
    for p in parts:
        p.tree = tree

    def evaluate():
        return sum(pc.full_traversal(p, one_by_one_pmatrices=True) for p in parts)
    lnl = evaluate()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lnl = evaluate()
    dt = (time.perf_counter() - t0) / reps
    upd = sum(len(tree.ops) * p.N * p.R for p in parts)
    out["C4_1gpu"] = {"ms_per_evaluation": dt * 1e3, "site_updates_per_s": upd / dt, "lnl": lnl,
                      "what": "2 x 250k-site DNA + 2 x 125k-site protein partitions, 100 taxa, linked "
                              "branch lengths, per-branch P-matrix calls, partitions evaluated in turn"}
    for p in parts:
        p.close()


def blo(lib, cfg, out, nsites=None):
    """one smoothing pass of Newton-Raphson branch-length optimisation over ALL branches
    through the C driver (pllhip_eval_optimize_branches = the reference's
    pllmod_opt_optimize_branch_lengths_local_multi pattern: per branch 1 sumtable,
    a few derivative scans, 1 P-matrix, up to 3 single-op CLV updates)"""
    S, R, ntips, N = pc.CONFIGS[cfg]
    N = nsites or N
    t = pc.Tree(ntips, 42, 43)
    ev = pc.Evaluation(lib, t.newick(), nparts=1)
    if S == 4:
        subst, freqs, alpha = pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841
    else:
        (subst, freqs), alpha = pc.protein_model(), 0.5
    inst = ev.add_partition(0, S, N, R, pc.random_codes(ntips, N, S), subst, freqs, alpha)
    with ev:
        if os.environ.get("PLLHIP_WORKLOAD_ATTACH", "1") != "0":
            ev.attach_comm(None)              # deferred results (what bench.py runs)
        l0 = ev.loglh()
        lib.lib.pllhip_synchronize(inst.p)
        ops0, pm0, d0 = ev.counters()
        n0 = ev.newton_iterations()
        t0 = time.perf_counter()
        l1 = ev.optimize_branches(1e-4, 10.0, 0.01, 1, -1)
        dt = time.perf_counter() - t0
        ops1, pm1, d1 = ev.counters()
        n1 = ev.newton_iterations()
        out[f"BLO_{cfg}_{N}"] = {"s_per_smoothing_pass": dt, "lnl_before": l0, "lnl_after": l1,
                                 "sumtable_scans": d1 - d0, "newton_iterations": n1 - n0,
                                 "single_op_updates": ops1 - ops0,
                                 "pmatrix_updates": pm1 - pm0, "branches": t.nedges,
                                 "us_per_scan_incl_everything": dt / max(1, d1 - d0) * 1e6,
                                 "us_per_derivative_call_incl_everything": dt / max(1, n1 - n0) * 1e6,
                                 "what": "pllhip_eval_optimize_branches(iters=1, radius=ALL), C driver; a "
                                         "derivative call = one Newton-Raphson iterate (the reference scans the "
                                         "sumtable once per iterate); scans evaluate up to 4 trial lengths"}


def blo_c4(lib, out, nsites=None):
    """the same smoothing pass over C4's four partitions under linked branch lengths (2 DNA + 2 protein partitions,
    BASELINE.json configs[3]; nsites: the configured total, e.g. 125 000 = the per-GPU slice of an 8-way split):
    the sums of the partitions' derivatives meet between two Newton iterates (src/optimize/pll_optimize.c:1223-1287)"""
    S, R, ntips, N = pc.CONFIGS["c4"]
    N = nsites or N
    t = pc.Tree(ntips, 42, 43)
    parts = [(4, 0.25), (4, 0.25), (20, 0.125), (20, 0.125)]
    ev = pc.Evaluation(lib, t.newick(), nparts=len(parts))
    insts = []
    for k, (s_, share) in enumerate(parts):
        n_ = max(1, int(round(N * share)))
        if s_ == 4:
            subst, freqs, alpha = pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841
        else:
            (subst, freqs), alpha = pc.protein_model(), 0.5
        insts.append(ev.add_partition(k, s_, n_, R, pc.random_codes(ntips, n_, s_, 44 + 101 * k), subst, freqs, alpha))
    with ev:
        l0 = ev.loglh()
        for i in insts:
            lib.lib.pllhip_synchronize(i.p)
        ops0, pm0, d0 = ev.counters()
        n0 = ev.newton_iterations()
        t0 = time.perf_counter()
        l1 = ev.optimize_branches(1e-4, 10.0, 0.01, 1, -1)
        dt = time.perf_counter() - t0
        ops1, pm1, d1 = ev.counters()
        n1 = ev.newton_iterations()
        out[f"BLO_c4_{N}"] = {"s_per_smoothing_pass": dt, "lnl_before": l0, "lnl_after": l1,
                              "newton_iterations": n1 - n0, "single_op_updates": ops1 - ops0,
                              "pmatrix_updates": pm1 - pm0, "branches": t.nedges, "partitions": len(parts),
                              "us_per_derivative_call_incl_everything": dt / max(1, n1 - n0) * 1e6,
                              # (4- and 20-state partitions: the loop over all of them in one launch, k_newton_multi)
                              "device_newton": os.environ.get("PLLHIP_EVAL_DEVICE_NEWTON", "1") != "0",
                              "what": "pllhip_eval_optimize_branches(iters=1, radius=ALL) over four partitions with linked "
                                      "branch lengths; a derivative call = one Newton-Raphson iterate of all partitions"}


def spr(lib, out, nsites=None, ntips=None, radius_max=5, thorough=False):
    """BASELINE config 5: codon GY94-like + G4, 50 taxa, one SPR round from a scrambled start"""
    S, R, taxa, N = pc.CONFIGS["c5"]
    N = nsites or N
    taxa = ntips or taxa
    truth = pc.Tree(taxa, 7, 8, brlen_range=(0.03, 0.25))
    start = pc.Tree(taxa, 11, 12, brlen_range=(0.05, 0.15))
    ev = pc.Evaluation(lib, start.newick(), nparts=1)
    subst, freqs = pc.codon_model()
    inst = ev.add_partition(0, S, N, R, pc.simulated_codes(truth, N, S), subst, freqs, 0.8)
    with ev:
        l0 = ev.loglh()
        lib.lib.pllhip_synchronize(inst.p)
        ops0, pm0, d0 = ev.counters()
        t0 = time.perf_counter()
        l1, st = ev.spr_round(radius_max=radius_max, ntopol_keep=5, thorough=thorough)
        dt = time.perf_counter() - t0
        ops1, pm1, d1 = ev.counters()
        out[f"SPR_c5_{taxa}x{N}_{'thorough' if thorough else 'fast'}"] = {
            "s_per_round": dt, "lnl_before": l0, "lnl_after": l1, "prunings": st.prunings,
            "insertions": st.insertions, "moves_applied": st.moves_applied, "rescored": st.rescored,
            "clv_ops": ops1 - ops0, "pmatrix_updates": pm1 - pm0, "derivative_calls": d1 - d0,
            "insertions_per_s": st.insertions / dt,
            "clv_site_updates_per_s": (ops1 - ops0) * N * R / dt,
            "what": f"pllhip_eval_spr_round(radius 1..{radius_max}, ntopol_keep 5), whole round incl. "
                    "branch-length optimisation of the remembered topologies"}


def alphabets(lib, out):
    """full traversals at the state counts of pll-modules' other model families (binary, genotype
    10 / 16 states: src/util/models_gt.c, multistate: src/util/models_mult.c): the 2..16-state
    matrix-core family.  PLLHIP_NO_S16=1 in the environment selects the generic kernels instead
    (one thread per output element), for the comparison."""
    for S, N in ((2, 1_000_000), (10, 500_000), (16, 500_000), (7, 500_000)):
        inst = pc.build_instance(lib, states=S, rate_cats=4, ntips=50, nsites=N, coded=True)
        with inst:
            pc.full_traversal(inst)
            lib.lib.pllhip_synchronize(inst.p)
            t0 = time.perf_counter()
            for _ in range(5):
                lnl = pc.full_traversal(inst)
            dt = (time.perf_counter() - t0) / 5
            name = lib.lib.pllhip_partials_kernel_name(inst.p).decode()
            # algorithmic bytes as SURVEY.md 8d counts them: 3 vectors of 8*S per (site, rate) and inner x inner op;
            # a coded tip child is 1 byte per site
            t = inst.tree
            nb = 0.0
            for op in t.ops:
                tips = (op[2] < t.ntips) + (op[5] < t.ntips)
                nb += N * 4 * 8.0 * S * (3 - tips) + N * tips + 12.0 * N
            out[f"ALPHABET_{S}states_{N}"] = {"kernel": name, "ms_per_traversal": dt * 1e3, "lnl": lnl,
                                              "site_updates_per_s": len(t.ops) * N * 4 / dt,
                                              "algorithmic_GBps": nb / dt / 1e9}


def alphabets32(lib, out):
    """the alphabets that left the generic kernels in round 3: 17 .. 32 states on the two-tile 2..32-state family,
    33 .. 64 states on the codon family with a run-time state count; 20 and 61 states next to them"""
    for S, N in ((17, 500_000), (20, 500_000), (24, 500_000), (28, 500_000), (32, 500_000), (48, 200_000),
                 (60, 200_000), (61, 200_000), (62, 200_000), (64, 200_000)):
        inst = pc.build_instance(lib, states=S, rate_cats=4, ntips=50, nsites=N, coded=True)
        with inst:
            pc.full_traversal(inst)
            lib.lib.pllhip_synchronize(inst.p)
            t0 = time.perf_counter()
            for _ in range(5):
                lnl = pc.full_traversal(inst)
            dt = (time.perf_counter() - t0) / 5
            t = inst.tree
            nb = 0.0
            for op in t.ops:
                tips = (op[2] < t.ntips) + (op[5] < t.ntips)
                nb += N * 4 * 8.0 * S * (3 - tips) + N * tips + 12.0 * N
            out[f"ALPHABET_{S}states_{N}"] = {"kernel": lib.lib.pllhip_partials_kernel_name(inst.p).decode(),
                                              "ms_per_traversal": dt * 1e3, "lnl": lnl,
                                              "site_updates_per_s": len(t.ops) * N * 4 / dt,
                                              "algorithmic_GBps": nb / dt / 1e9}


def alphabets_repeats(lib, out):
    """PLL_ATTRIB_SITE_REPEATS in the 2 .. 32-state family: full traversals with and without the attribute, on iid
    uniform states and on an alignment simulated along the tree (seed 45); identical lnL asked for"""
    for S, N in ((2, 1_000_000), (5, 1_000_000), (10, 500_000), (16, 500_000), (24, 500_000)):
        tree = pc.Tree(50, 42, 43, brlen_range=(0.01, 0.12))
        for data in ("random", "simulated"):
            codes = pc.simulated_codes(tree, N, S, seed=45) if data == "simulated" else None
            res = {}
            for attr in (0, pc.PLL_ATTRIB_SITE_REPEATS):
                inst = pc.build_instance(lib, states=S, rate_cats=4, ntips=50, nsites=N, coded=True, tree=tree,
                                         attributes=attr, codes=codes)
                inst.tree = tree
                with inst:
                    pc.full_traversal(inst)
                    pc.full_traversal(inst)
                    lib.lib.pllhip_synchronize(inst.p)
                    t0 = time.perf_counter()
                    for _ in range(5):
                        lnl = pc.full_traversal(inst)
                    dt = (time.perf_counter() - t0) / 5
                    st = inst.repeat_stats()
                    res["on" if attr else "off"] = {"ms_per_traversal": dt * 1e3, "lnl": lnl,
                                                    "class_operations_per_traversal": st.cherries // 7 if attr else 0}
            assert res["on"]["lnl"] == res["off"]["lnl"], (S, data, res)
            out[f"REPEATS_{S}states_{N}_{data}"] = {"ms_attribute_off": res["off"]["ms_per_traversal"],
                                                    "ms_attribute_on": res["on"]["ms_per_traversal"],
                                                    "class_operations_per_traversal": res["on"]["class_operations_per_traversal"],
                                                    "operations": len(tree.ops), "lnl": res["on"]["lnl"]}


def main():
    lib = pc.PllLib(pc.PRODUCT_LIB)
    out = {}
    which = sys.argv[1:] or ["w2", "w3", "c4", "blo"]
    if "alphabets_repeats" in which:
        alphabets_repeats(lib, out)
    if "alphabets" in which:
        alphabets(lib, out)
    if "alphabets32" in which:
        alphabets32(lib, out)
    if "blo125" in which:
        blo(lib, "c3", out, nsites=125_000)
    if "blo_c2" in which:
        blo(lib, "c2", out, nsites=125_000)
    if "blo" in which:
        blo(lib, "c3", out)
        blo(lib, "c2", out)
        blo(lib, "c3", out, nsites=125_000)
    if "blo_c4_125" in which:
        blo_c4(lib, out, nsites=125_000)
    if "blo_c4" in which:
        blo_c4(lib, out, nsites=125_000)
        blo_c4(lib, out)
    if "w2" in which:
        w2(lib, "c3", out)
        w2(lib, "c2", out)
    if "w3" in which:
        w3(lib, "c3", out)
        w3(lib, "c2", out)
        w3(lib, "c3", out, nsites=125_000)
    if "c4" in which:
        c4(lib, out)
    if "spr" in which or "spr200" in which:
        spr(lib, out)
    if "spr" in which or "spr25" in which:
        spr(lib, out, nsites=25_000)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
