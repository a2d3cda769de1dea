"""The C evaluation driver (include/pllhip_eval.h): treeinfo-style incremental
likelihood and Newton-Raphson branch-length optimisation over several
partitions.  CPU tests run it on the oracle; GPU tests run it on the HIP engine
and compare with the oracle (same C code, different library underneath).
tests/test_dropin_modules.py additionally checks it against the reference's own
treeinfo / optimiser on identical data."""
import ctypes as C
import os

import numpy as np
import pytest

import pllhip_ctypes as pc


def build(lib, ntips=14, flags=0, sizes=(400, 150), coded=True):
    t = pc.Tree(ntips, 42, 43)
    ev = pc.Evaluation(lib, t.newick(), flags=flags, nparts=2)
    ev.add_partition(0, 4, sizes[0], 4, pc.random_codes(ntips, sizes[0], 4), pc.DNA_GTR_RATES,
                     pc.DNA_FREQS, 0.7, coded=coded)
    r, f = pc.protein_model()
    ev.add_partition(1, 20, sizes[1], 4, pc.random_codes(ntips, sizes[1], 20, seed=99), r, f, 0.5,
                     coded=coded)
    ev.pytree = t
    return ev


def reference_lnl(lib, ev):
    """the same two partitions evaluated without the driver (plain op lists)"""
    t = ev.pytree
    total = 0.0
    for states, n, seed, (subst, freqs), alpha in ((4, ev.parts[0].N, 44, (pc.DNA_GTR_RATES, pc.DNA_FREQS), 0.7),
                                                   (20, ev.parts[1].N, 99, pc.protein_model(), 0.5)):
        inst = pc.Instance(lib, t.ntips, states, n, 4, attributes=pc.PLL_ATTRIB_PATTERN_TIP)
        inst.set_model(subst, freqs, lib.gamma_cats(alpha, 4))
        codes = pc.random_codes(t.ntips, n, states, seed=seed)
        cmap = pc.state_charmap(states)
        for k in range(t.ntips):
            inst.set_tip_states(k, cmap, (codes[k] + 48).tobytes())
        inst.tree = t
        total += pc.full_traversal(inst)
        inst.close()
    return total


def _walk_records(ev):
    tr = ev.utree.contents
    for i in range(tr.tip_count + tr.inner_count):
        n = tr.nodes[i]
        s = n
        while True:
            yield s
            if not s.contents.next:
                break
            s = s.contents.next
            if C.addressof(s.contents) == C.addressof(n.contents):
                break


def check_driver(lib, other=None):
    with build(lib) as ev:
        full = ev.loglh()
        assert abs(full - reference_lnl(lib, ev)) < 1e-8 * abs(full)
        ops0, pm0, _ = ev.counters()
        assert (ops0, pm0) == (12, 25)                       # n-2 ops, 2n-3 matrices
        # nothing invalid: an incremental evaluation does no work
        assert ev.loglh(True) == full and ev.counters()[:2] == (ops0, pm0)
        # invalidate an inner CLV and the root above it (as in the reference, a valid
        # CLV shields everything below it: src/tree/treeinfo.c:38-61): only those
        # two operations are redone
        root = lib.lib.pllhip_eval_root(ev.ev)
        victim = root.contents.next.contents.back
        if not victim.contents.next:
            victim = root.contents.next.contents.next.contents.back
        lib.lib.pllhip_eval_invalidate_clv(ev.ev, victim)
        assert ev.loglh(True) == full and ev.counters()[0] == ops0      # shielded by the valid root
        lib.lib.pllhip_eval_invalidate_clv(ev.ev, root)
        assert abs(ev.loglh(True) - full) < 1e-9 * abs(full)
        assert ev.counters()[0] - ops0 == 2
        # re-rooting at every inner record gives the same likelihood, incrementally
        for rec in _walk_records(ev):
            if rec.contents.next:
                assert lib.lib.pllhip_eval_set_root(ev.ev, rec)
                assert abs(ev.loglh(True) - full) < 1e-9 * abs(full)
        # a branch-length change is picked up
        lib.lib.pllhip_eval_set_branch_length(ev.ev, lib.lib.pllhip_eval_root(ev.ev), 0.7)
        changed = ev.loglh(True)
        assert abs(changed - full) > 1e-3
        assert abs(changed - ev.loglh(False)) < 1e-9 * abs(changed)
        # Newton-Raphson over all branches: improves, and the result is consistent
        # with a fresh full evaluation
        opt = ev.optimize_branches(1e-4, 10.0, 0.01, 8, -1)
        assert opt > changed + 1.0
        assert abs(ev.loglh(False) - opt) < 1e-6 * abs(opt)
        assert ev.counters()[2] > 25
        return full, changed, opt


def test_driver_on_oracle(oracle):
    check_driver(oracle)


def test_per_branch_pmatrix_flag_gives_identical_results(oracle):
    with build(oracle) as a, build(oracle, flags=1) as b:
        assert a.loglh() == b.loglh()
        assert a.optimize_branches() == b.optimize_branches()


def test_remote_partitions_and_reduce_callback(oracle):
    """two 'workers' in one process: each owns one partition (the other slot is
    NULL, src/tree/treeinfo.c:1024-1029) and a reduce callback adds the other
    worker's share; both must see the full-alignment likelihood"""
    with build(oracle) as whole:
        want = whole.loglh()
        per_part = [p for p in whole.lib.lib.pllhip_eval_loglh.argtypes]  # noqa: F841 (keep lib alive)
    t = pc.Tree(14, 42, 43)
    shares = {}

    def make(owner):
        ev = pc.Evaluation(oracle, t.newick(), nparts=2)
        if owner == 0:
            ev.add_partition(0, 4, 400, 4, pc.random_codes(14, 400, 4), pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.7)
            oracle.lib.pllhip_eval_set_partition(ev.ev, 1, None, None)
        else:
            r, f = pc.protein_model()
            oracle.lib.pllhip_eval_set_partition(ev.ev, 0, None, None)
            ev.add_partition(1, 20, 150, 4, pc.random_codes(14, 150, 20, seed=99), r, f, 0.5)
        return ev
    a, b = make(0), make(1)
    # first pass without a callback: each worker sees only its own share
    shares[0], shares[1] = a.loglh(), b.loglh()
    assert abs(shares[0] + shares[1] - want) < 1e-9 * abs(want)

    def cb_for(other_share, slot):
        def cb(ctx, data, n, op):
            assert n == 2 and op == 0
            data[slot] += other_share
        return pc.REDUCE_CB(cb)
    cba, cbb = cb_for(shares[1], 1), cb_for(shares[0], 0)
    oracle.lib.pllhip_eval_set_parallel_context(a.ev, None, C.cast(cba, C.c_void_p))
    oracle.lib.pllhip_eval_set_parallel_context(b.ev, None, C.cast(cbb, C.c_void_p))
    assert abs(a.loglh() - want) < 1e-9 * abs(want)
    assert abs(b.loglh() - want) < 1e-9 * abs(want)
    a.close()
    b.close()


@pytest.mark.gpu
def test_driver_on_gpu_matches_oracle(product, oracle):
    g = check_driver(product)
    c = check_driver(oracle)
    for x, y in zip(g, c):
        assert abs(x - y) < 1e-7 * abs(y)


@pytest.mark.gpu
def test_driver_counters_reach_the_engine(product):
    with build(product, flags=1) as ev:
        ev.loglh()
        c = ev.parts[1].counters()
        assert c.pmatrix_updates == 25 and c.pmatrix_launches == 1   # 25 calls, one launch
        assert c.partial_ops == 12 and c.partial_launches < 12       # level-scheduled


# C4 (BASELINE.json configs[3]): two DNA + two protein partitions under one 100-taxon tree with linked
# branch lengths, evaluated through the driver as pllmod_treeinfo does (src/tree/treeinfo.c:1020-1056)
C4_SHAPE = [(4, 250_000), (4, 250_000), (20, 125_000), (20, 125_000)]


def _c4_evaluation(lib, tree, sizes, tile_codes=None, tile=0):
    ev = pc.Evaluation(lib, tree.newick(), flags=1, nparts=len(C4_SHAPE))
    codes_out = []
    for k, ((states, _), n) in enumerate(zip(C4_SHAPE, sizes)):
        subst, freqs, alpha = (pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841) if states == 4 else (*pc.protein_model(), 0.5)
        codes = pc.random_codes(tree.ntips, n, states, 44 + 101 * k) if tile_codes is None else \
            np.tile(tile_codes[k], (1, n // tile))
        codes_out.append(codes)
        ev.add_partition(k, states, n, 4, codes, subst, freqs, alpha, coded=True)
    return ev, codes_out


@pytest.mark.gpu
def test_c4_shape_at_full_size_through_tiling(product, oracle):
    """C4 at its own shape -- 4 partitions, 100 taxa, 2 x 250 k DNA + 2 x 125 k protein sites -- is out of the
    oracle's reach in seconds; an alignment made of K copies of a 500-site tile per partition is not:
    lnL(full) = sum_p K_p lnL_p(tile), per-site lnL periodic in the tile, incremental re-evaluation after a
    branch-length change agrees with the tile as well"""
    tile = 500
    t = pc.Tree(100, 42, 43)
    ev_o, codes = _c4_evaluation(oracle, t, [tile] * 4)
    with ev_o:
        l_tile = ev_o.loglh()
        per_o = [ev_o.persite_lnl(k)[1] for k in range(4)]
    ev_t, _ = _c4_evaluation(product, t, [tile] * 4, tile_codes=codes, tile=tile)
    with ev_t:
        l_ref = ev_t.loglh()
        per_g = [ev_t.persite_lnl(k)[1] for k in range(4)]
    assert abs(l_ref - l_tile) <= 2e-9 * 4 * tile
    for a, b in zip(per_g, per_o):
        assert np.abs(a - b).max() < 1e-9
    sizes = [n for _, n in C4_SHAPE]
    want = sum((n // tile) * per_o[k].sum() for k, n in enumerate(sizes))
    ev, _ = _c4_evaluation(product, t, sizes, tile_codes=codes, tile=tile)
    with ev:
        l_full = ev.loglh()
        assert abs(l_full - want) <= 1e-10 * abs(want)
        for k, n in enumerate(sizes):
            ps = ev.persite_lnl(k)[1]
            assert np.array_equal(ps.reshape(n // tile, tile), np.broadcast_to(ps[:tile], (n // tile, tile)))
        # incremental: one branch changes, the driver recomputes the path to the root only
        rec = next(r for r in ev.records() if r.contents.next)
        ev.L.pllhip_eval_set_branch_length(ev.ev, rec, 0.31)
        l_inc = ev.loglh(incremental=True)
    ev_o2, _ = _c4_evaluation(oracle, t, [tile] * 4)
    with ev_o2:
        ev_o2.loglh()
        rec = next(r for r in ev_o2.records() if r.contents.next)
        ev_o2.L.pllhip_eval_set_branch_length(ev_o2.ev, rec, 0.31)
        ev_o2.loglh(incremental=True)
        per2 = [ev_o2.persite_lnl(k)[1] for k in range(4)]
    want2 = sum((n // tile) * per2[k].sum() for k, n in enumerate(sizes))
    assert abs(l_inc - want2) <= 1e-9 * abs(want2) and abs(l_inc - l_full) > 1.0


# ---------------------------------------------------------------------------
# SPR round (pllhip_eval_spr_round; tests/test_dropin_modules.py pins it against
# the reference's pllmod_algo_spr_round where the reference is present)
# ---------------------------------------------------------------------------
def build_search(lib, states, ntips=12, nsites=300, true_seed=7, start_seed=11, model=None):
    """alignment simulated on one tree, evaluation started on another"""
    truth = pc.Tree(ntips, true_seed, true_seed + 1, brlen_range=(0.03, 0.25))
    start = pc.Tree(ntips, start_seed, start_seed + 1, brlen_range=(0.05, 0.15))
    ev = pc.Evaluation(lib, start.newick(), nparts=1)
    if model is None:
        model = {4: (pc.DNA_GTR_RATES, pc.DNA_FREQS), 20: pc.protein_model(), 61: pc.codon_model()}[states]
    ev.add_partition(0, states, nsites, 4, pc.simulated_codes(truth, nsites, states), model[0], model[1], 0.8)
    return ev


def run_rounds(lib, states, thorough, rounds=2, **kw):
    with build_search(lib, states, **kw) as ev:
        cut = pc.SprCutoff(0.0, 1e30, 0.0, 0)
        first = ev.loglh()
        trace = []
        for _ in range(rounds):
            lnl, st = ev.spr_round(radius_max=4, ntopol_keep=4, thorough=thorough, cutoff=cut)
            assert st.lnl_start <= lnl + 1e-9
            assert abs(ev.loglh() - lnl) < 1e-6                 # restored tree re-evaluates to the result
            trace.append((lnl, st.prunings, st.insertions, st.moves_applied, st.rescored,
                          tuple(st.log_prune[:st.log_count]), tuple(st.log_regraft[:st.log_count]),
                          cut.lh_dec_count, cut.lh_dec_sum))
        return first, trace, ev.newick()


@pytest.mark.parametrize("states,thorough", [(4, False), (4, True), (20, False)])
def test_spr_round_improves_and_is_deterministic(oracle, states, thorough):
    first, trace, nwk = run_rounds(oracle, states, thorough)
    assert trace[0][0] > first + 1.0                            # the scrambled start is far from optimal
    assert trace[0][3] > 0 and trace[0][2] > trace[0][1]        # moves applied; several insertions per pruning
    assert trace[1][0] >= trace[0][0] - 1e-6
    again = run_rounds(oracle, states, thorough)
    assert again == (first, trace, nwk)                         # bit-identical rerun (algo_search.c:1453)


def test_spr_round_rejects_bad_parameters(oracle):
    with build_search(oracle, 4) as ev:
        with pytest.raises(RuntimeError):
            ev.spr_round(radius_min=0)
        with pytest.raises(RuntimeError):
            ev.spr_round(radius_min=3, radius_max=2)


@pytest.mark.gpu
@pytest.mark.parametrize("states,thorough", [(4, False), (20, False), (20, True), (61, False)])
def test_spr_round_on_gpu_makes_the_oracle_s_moves(product, oracle, states, thorough):
    kw = dict(ntips=10, nsites=200) if states == 61 else {}
    g = run_rounds(product, states, thorough, **kw)
    c = run_rounds(oracle, states, thorough, **kw)
    assert abs(g[0] - c[0]) < 1e-8 * abs(c[0])
    for a, b in zip(g[1], c[1]):
        assert a[1:8] == b[1:8]                                 # same scan, same accepted moves, in order
        assert abs(a[0] - b[0]) < 1e-6 and abs(a[8] - b[8]) < 1e-6 * max(1.0, abs(b[8]))


def _random_walk(lib, steps, seed, ntips=40):
    """random re-rootings, CLV invalidations and branch-length changes, evaluated
    incrementally: arbitrary partial operation lists reach pll_update_partials"""
    rng = np.random.default_rng(seed)
    out = []
    with build(lib, ntips=ntips, sizes=(333, 97)) as ev:
        ev.loglh()
        recs = [r for r in _walk_records(ev)]
        inner = [r for r in recs if r.contents.next]
        for _ in range(steps):
            what = rng.integers(0, 4)
            if what == 0:
                assert lib.lib.pllhip_eval_set_root(ev.ev, inner[rng.integers(len(inner))])
            elif what == 1:
                for k in rng.integers(0, len(inner), size=rng.integers(1, 6)):
                    lib.lib.pllhip_eval_invalidate_clv(ev.ev, inner[k])
            elif what == 2:
                rec = recs[rng.integers(len(recs))]
                lib.lib.pllhip_eval_set_branch_length(ev.ev, rec, float(rng.uniform(0.001, 1.5)))
            else:
                lib.lib.pllhip_eval_invalidate_all(ev.ev)
            out.append(ev.loglh(True))
        out.append(ev.loglh(False))
    return out


def test_random_incremental_walk_is_consistent(oracle):
    vals = _random_walk(oracle, 40, 5, ntips=20)
    assert abs(vals[-1] - vals[-2]) < 1e-9 * abs(vals[-1])        # incremental == full at the end


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_incremental_walk_on_gpu_matches_oracle(product, oracle, seed):
    g = _random_walk(product, 60, seed)
    c = _random_walk(oracle, 60, seed)
    for x, y in zip(g, c):
        assert abs(x - y) < 1e-9 * abs(y)
    assert abs(g[-1] - g[-2]) < 1e-10 * abs(g[-1])


def check_speculative_newton(lib):
    """Newton-Raphson with several trial lengths per sumtable scan (the clamped outcomes of the
    step rule) makes the same iterates and returns the same numbers, bit for bit, as one scan
    per iteration -- with fewer scans"""
    runs = []
    for flags in (4, 2):                 # PLLHIP_EVAL_ALWAYS_SPECULATE, PLLHIP_EVAL_NO_SPECULATION
        with build(lib, flags=flags) as ev:
            ev.loglh()
            opt = ev.optimize_branches(1e-4, 10.0, 0.01, 8, -1)
            runs.append((opt, ev.newick(), ev.counters()[2], ev.newton_iterations()))
    (opt_s, tree_s, scans_s, iters_s), (opt_1, tree_1, scans_1, iters_1) = runs
    assert opt_s == opt_1 and tree_s == tree_1 and iters_s == iters_1
    assert scans_1 == iters_1 and scans_s < scans_1
    return scans_s, scans_1


def test_speculative_newton_on_oracle(oracle):
    check_speculative_newton(oracle)


def test_transient_mode_is_harmless_without_an_engine_that_has_it(oracle):
    """pllhip_eval_set_transient on the CPU oracle (which stores every vector and exports no pllhip_set_transient:
    the driver binds those symbols weakly): the same numbers in every mode"""
    out = []
    for mode in (0, 1, 2):
        with build(oracle) as ev:
            ev.set_transient(mode)
            seq = [ev.loglh(), ev.loglh(), ev.optimize_branches(1e-4, 10.0, 0.01, 2, -1), ev.loglh(incremental=True), ev.loglh()]
            out.append(seq)
    assert out[0] == out[1] == out[2]


def test_multi_length_derivatives_on_oracle(oracle):
    inst = pc.build_instance(oracle, states=20, rate_cats=4, ntips=8, nsites=200, coded=True)
    with inst:
        pc.full_traversal(inst)
        t = inst.tree
        st = inst.alloc_sumtable()
        a = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        inst.update_sumtable(*a, st)
        ts = [0.3, 1e-4, 0.05, 2.0, 0.3]
        df, ddf = inst.derivatives_multi(a[2], a[3], ts, st)
        for k, x in enumerate(ts):
            assert (df[k], ddf[k]) == inst.derivatives(a[2], a[3], x, st)
        inst.free_sumtable(st)


def run_linkage(lib, linkage, attach=False):
    with build(lib, ntips=10) as ev:
        if attach:
            ev.attach_comm(None)
        ev.set_linkage(linkage, [0.6, 1.8])
        lnl = ev.loglh()
        opt = ev.optimize_branches(1e-4, 10.0, 0.01, 6, -1)
        return lnl, opt, ev.loglh(), ev.partition_tree_length(0), ev.partition_tree_length(1)


@pytest.mark.parametrize("linkage", [1, 2])
def test_scaled_and_unlinked_lengths_on_oracle(oracle, linkage):
    """per-partition scalers / per-partition lengths (tests/test_dropin_modules.py pins both modes
    against the reference's treeinfo): optimisation improves, re-evaluation agrees, and the two
    partitions end up with different effective tree lengths"""
    lnl, opt, again, t0, t1 = run_linkage(oracle, linkage)
    assert opt > lnl + 1.0 and abs(again - opt) < 1e-6 * abs(opt)
    assert abs(t0 - t1) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("linkage", [1, 2])
def test_scaled_and_unlinked_lengths_on_gpu_match_oracle(product, oracle, linkage):
    g = run_linkage(product, linkage)
    c = run_linkage(oracle, linkage)
    f = run_linkage(product, linkage, attach=True)      # deferred results: same path, one wait per round
    for x, y, z in zip(g, c, f):
        assert abs(x - y) < 1e-7 * max(1.0, abs(y))
        assert abs(z - x) < 1e-9 * max(1.0, abs(x))


# ---------------------------------------------------------------------------
# Newton-Raphson on the device (include/pllhip.h, pllhip_newton_branch)
# ---------------------------------------------------------------------------
def _host_newton(deriv, x, bl_min, bl_max, tol, max_newton):
    """the step rule of newton() (csrc/host/pllhip_eval.c; reference: src/optimize/opt_algorithms.c:133-261)
    in Python floats (IEEE doubles, no contraction): the iterate after every scan"""
    dxmax = bl_max / max_newton
    xl, xh = bl_min, bl_max
    x = max(min(x, bl_max), bl_min)
    trail = []
    it = 0
    while True:
        if it > max_newton:
            raise OverflowError("Exceeded maximum number of iterations")
        it += 1
        f, df = deriv(x)
        if df > 0.0:
            if abs(f) < tol:
                trail.append(x)
                return x, trail
            if f < 0.0:
                xl = x
            else:
                xh = x
            dx = -f / df
        else:
            dx = -f / abs(df)
        dx = max(min(dx, dxmax), -dxmax)
        if x + dx < xl:
            dx = xl - x
        if x + dx > xh:
            dx = xh - x
        if abs(dx) < tol:
            trail.append(x)
            return x, trail
        x += dx
        x = max(min(x, bl_max), bl_min)
        trail.append(x)


@pytest.mark.gpu
@pytest.mark.parametrize("states,nsites,pinv", [(20, 5000, 0.0), (20, 333, 0.1), (61, 700, 0.0), (10, 2000, 0.0),
                                                (24, 1500, 0.0), (62, 300, 0.0), (2, 4000, 0.0),
                                                (4, 5000, 0.0), (4, 777, 0.2), (4, 200_000, 0.0), (20, 120_000, 0.0), (10, 150_000, 0.0)])
def test_device_newton_makes_the_host_loop_s_iterates(product, states, nsites, pinv):
    """one launch runs the whole Newton-Raphson loop of a branch: every iterate is, bit for bit, the one the
    host loop reaches by calling pll_compute_likelihood_derivatives once per iterate"""
    with pc.build_instance(product, states=states, rate_cats=4, ntips=9, nsites=nsites, coded=True, pinv=pinv) as a:
        pc.full_traversal(a)
        t = a.tree
        st = a.alloc_sumtable()
        # every edge of the tree would need re-rooting; the root edge and, after re-rooting, two more
        for root_edge in (t.root_matrix, 0, t.nedges // 2):
            t2 = pc.Tree(9, 42, 43)
            t2.set_root_edge(root_edge)
            a.tree = t2
            pc.full_traversal(a)
            if t2.root_b < t2.ntips and t2.root_a < t2.ntips:
                continue
            sa, sb = t2.scaler_of(t2.root_a), t2.scaler_of(t2.root_b)
            a.update_sumtable(t2.root_a, t2.root_b, sa, sb, st)
            for start in (float(t2.brlens[root_edge]), 1e-4, 5.0):
                try:
                    want_x, want_trail = _host_newton(lambda x: a.derivatives(sa, sb, x, st), start, 1e-4, 10.0, 1e-5, 32)
                except OverflowError:
                    with pytest.raises(RuntimeError, match="910"):      # the reference's failure mode, on both sides
                        a.newton_branch(sa, sb, st, start, 1e-4, 10.0, 1e-5, 32)
                    continue
                got_x, its, trail = a.newton_branch(sa, sb, st, start, 1e-4, 10.0, 1e-5, 32)
                assert its == len(want_trail), (states, root_edge, start, its, len(want_trail))
                assert list(trail) == want_trail
                assert got_x == want_x
        a.free_sumtable(st)


def _needs_a_queue_per_stream(states=()):
    """Partitions of the 4-state and the matrix-core families run the loop over several partitions in ONE launch
    (k_newton_multi).  The other form -- one launch per partition -- runs only in a process that gave every partition
    stream a hardware queue of its own before its first HIP call (pll_core.hip, newton_multi_enabled):
    tests/test_00_forced_modes.py runs these tests once more in such a child process, with the one-launch form off"""
    one_launch = os.environ.get("PLLHIP_NEWTON_ONE_LAUNCH", "1") != "0" and all(2 <= s <= 64 for s in states)
    if not one_launch and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 8:
        pytest.skip("GPU_MAX_HW_QUEUES >= 8 needed before the first HIP call (run by tests/test_00_forced_modes.py)")


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [293, 292, 64])
def test_device_newton_with_a_capped_reduction_grid(blocks):
    """150 000 sites x 4 rates at 4 states: with 293 workgroups of 1 024 columns a thread makes two trips over the sumtable
    (the loop keeps it in registers: every iterate is the host loop's), with 292 three and with 64 ten: the residency
    decision has to be made for the grid that is launched -- PLLHIP_REDUCE_BLOCKS caps it --, and then the entry point
    declines (PLLHIP_ERROR_NEWTON_UNSUPPORTED: the callers iterate from the host) instead of running a loop whose threads
    make more trips than they hold registers for (tests/_newton_trips_worker.py: the cap is read when the library is loaded)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "_newton_trips_worker.py"), "150000", str(blocks)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert any(h is not None for h in got["host"])
    if blocks >= 293:
        assert got["device"] == [h if h is not None else 910 for h in got["host"]]
    else:
        assert got["device"] == [912] * 3


@pytest.mark.gpu
@pytest.mark.parametrize("spec,scalers", [
    ([(20, 3000), (20, 1700)], None),
    ([(4, 6000), (4, 2500), (20, 1500), (20, 900)], None),                 # C4's mix: two kernel families in one loop
    ([(4, 6000), (4, 2500), (20, 1500), (20, 900)], [1.0, 0.7, 1.9, 1.2]),   # scaled branch lengths: chain rule s, s^2
    ([(61, 300), (20, 800), (10, 1200), (4, 50_000)], [0.8, 1.0, 1.3, 1.1]),
    ([(61, 300), (20, 800), (62, 200), (4, 50_000)], [0.8, 1.0, 1.3, 1.1]),         # every family of the one-launch form
    ([(20, 60_000), (20, 50_000)], None),                                   # register-resident instances side by side
])
def test_device_newton_over_several_partitions(product, spec, scalers):
    """pllhip_newton_branch_multi: partitions that share a branch length run ONE loop on the device; every iterate is
    the host loop's -- f = sum s_p f_p(s_p x), f' = sum s_p^2 f'_p(s_p x) added in partition order
    (src/optimize/pll_optimize.c:1223-1287) -- bit for bit"""
    _needs_a_queue_per_stream([s_ for s_, _ in spec])
    tree = pc.Tree(9, 42, 43)
    insts = [pc.build_instance(product, states=s_, rate_cats=4, ntips=9, nsites=n_, coded=True, tree=tree, seed_shift=k)
             for k, (s_, n_) in enumerate(spec)]
    sc = scalers or [1.0] * len(insts)
    try:
        sts = []
        for a in insts:
            a.tree = tree
            pc.full_traversal(a)
            sts.append(a.alloc_sumtable())
        for root_edge in (tree.root_matrix, 2):
            t2 = pc.Tree(9, 42, 43)
            t2.set_root_edge(root_edge)
            if t2.root_b < t2.ntips and t2.root_a < t2.ntips:
                continue
            sa, sb = t2.scaler_of(t2.root_a), t2.scaler_of(t2.root_b)
            for a, st in zip(insts, sts):
                a.tree = t2
                pc.full_traversal(a)
                a.update_sumtable(t2.root_a, t2.root_b, sa, sb, st)

            def deriv(x):
                f = df = 0.0
                for a, st, s_ in zip(insts, sts, sc):
                    fp, dfp = a.derivatives(sa, sb, s_ * x, st)
                    f += s_ * fp
                    df += s_ * s_ * dfp
                return f, df
            for start in (float(t2.brlens[root_edge]), 1e-4, 3.0):
                try:
                    want_x, want_trail = _host_newton(deriv, start, 1e-4, 10.0, 1e-5, 32)
                except OverflowError:
                    with pytest.raises(RuntimeError, match="910"):
                        pc.newton_branch_multi(product, insts, sa, sb, sts, scalers, start, 1e-4, 10.0, 1e-5, 32)
                    continue
                got_x, its, trail = pc.newton_branch_multi(product, insts, sa, sb, sts, scalers, start, 1e-4, 10.0, 1e-5, 32)
                assert its == len(want_trail), (root_edge, start, its, len(want_trail))
                assert list(trail) == want_trail
                assert got_x == want_x
        for a, st in zip(insts, sts):
            a.free_sumtable(st)
    finally:
        for a in insts:
            a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("linkage", [0, 1])
def test_driver_with_device_newton_over_partitions(product, linkage):
    """pllhip_eval_optimize_branches over C4's mix of partitions (linked, and scaled branch lengths): the device loop
    and the host loop give the same tree, the same lnL and the same number of Newton iterations"""
    _needs_a_queue_per_stream([4, 20])
    tree = pc.Tree(12, 42, 43)
    out, launches = [], []
    for flag in ("0", "1"):
        os.environ["PLLHIP_EVAL_DEVICE_NEWTON"] = flag
        try:
            with pc.Evaluation(product, tree.newick(), nparts=4) as ev:
                for k, (s_, n_) in enumerate([(4, 3000), (4, 2000), (20, 900), (20, 700)]):
                    subst, freqs = (pc.protein_model() if s_ == 20 else (pc.DNA_GTR_RATES, pc.DNA_FREQS))
                    ev.add_partition(k, s_, n_, 4, pc.simulated_codes(tree, n_, s_, 45 + k), subst, freqs, 0.7)
                if linkage:
                    ev.set_linkage(1, [1.0, 0.8, 1.5, 1.1])
                l0 = ev.loglh()
                l1 = ev.optimize_branches(1e-4, 10.0, 0.01, 3, -1)
                out.append((l0, l1, ev.newick(), ev.newton_iterations()))
                launches.append(sum(p_.counters().derivative_calls for p_ in ev.parts))
        finally:
            del os.environ["PLLHIP_EVAL_DEVICE_NEWTON"]
    assert out[0] == out[1]
    assert launches[1] < launches[0] // 2        # one launch per partition and branch instead of one per iterate


@pytest.mark.gpu
def test_device_newton_says_when_it_cannot(product):
    with pc.build_instance(product, states=4, rate_cats=4, ntips=6, nsites=500, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        st = a.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b), st)
        with pytest.raises(RuntimeError, match="912"):       # no iterations allowed: nothing for the loop to do
            a.newton_branch(t.scaler_of(t.root_a), t.scaler_of(t.root_b), st, 0.1, 1e-4, 10.0, 1e-5, 0)
        a.free_sumtable(st)
    with pc.build_instance(product, states=20, rate_cats=4, ntips=6, nsites=500, coded=True) as a:
        pc.full_traversal(a)
        t = a.tree
        st = a.alloc_sumtable()
        a.update_sumtable(t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b), st)
        with pytest.raises(RuntimeError, match="910"):       # an iteration limit the loop cannot meet
            a.newton_branch(t.scaler_of(t.root_a), t.scaler_of(t.root_b), st, 9.0, 1e-4, 10.0, 1e-12, 1)
        a.free_sumtable(st)


@pytest.mark.gpu
@pytest.mark.parametrize("states", [20, 61, 10, 4])
def test_driver_with_device_newton_equals_driver_with_host_loop(product, states):
    """pllhip_eval_optimize_branches on one partition: the device loop and the host loop give the same tree,
    the same lnL and the same number of Newton iterations"""
    import os
    out = []
    for flag in ("0", "1"):
        os.environ["PLLHIP_EVAL_DEVICE_NEWTON"] = flag
        try:
            model = None
            if states == 10:
                model = (0.5 + pc.uniform01(156, 45), np.full(10, 0.1))
            with build_search(product, states, ntips=12, nsites=400 if states < 61 else 150, model=model) as ev:
                l0 = ev.loglh()
                l1 = ev.optimize_branches(1e-4, 10.0, 0.01, 3, -1)
                out.append((l0, l1, ev.newick(), ev.newton_iterations()))
        finally:
            del os.environ["PLLHIP_EVAL_DEVICE_NEWTON"]
    assert out[0] == out[1]
