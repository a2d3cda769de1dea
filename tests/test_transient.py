"""Evaluate-only traversals (include/pllhip.h, pllhip_set_transient): while the mode is on a resident schedule hands
the vectors inside its operation chains on in registers without storing them.  The model-parameter optimisers of the
reference evaluate the whole tree after every parameter change (src/algorithm/algo_callback.c:338, 465, 568, 678;
src/optimize/opt_algorithms.c:734-773), so those vectors are recomputed before anybody reads them.  Whatever a
caller CAN observe must be what it observes with the mode off, bit for bit -- a vector that was not stored is
recomputed for its first reader, or before one of its inputs changes -- and must agree with the CPU oracle."""
import numpy as np
import pytest

import common
import pllhip_ctypes as pc
from test_gpu_parity import lnl_close, site_err, REL_CLV, CLV_SITE_61
from test_site_repeats import _everything, _same

# (with every partition forced to compute per class of sites the evaluate-only mode is off: kernels_repeats.hpp)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(common.FORCED_REPEATS, reason="evaluate-only traversals are off under site repeats")]
NONE = pc.PLL_SCALE_BUFFER_NONE

FAMILIES = [(20, 4), (4, 4), (10, 4), (2, 4), (24, 4), (4, 2), (20, 2)]


def _build(lib, states, rate_cats, ntips, nsites, tree=None, transient=False, **kw):
    inst = pc.build_instance(lib, states=states, rate_cats=rate_cats, ntips=ntips, nsites=nsites, coded=True, tree=tree, **kw)
    if transient:
        inst.set_transient(True)
    return inst


@pytest.mark.parametrize("states,rate_cats", FAMILIES)
@pytest.mark.parametrize("ntips,nsites", [(14, 1031), (60, 4100), (120, 700)])
def test_transient_traversals_change_nothing_a_caller_can_see(product, oracle, states, rate_cats, ntips, nsites):
    tree = pc.Tree(ntips, 42, 43)
    with _build(product, states, rate_cats, ntips, nsites, tree, True) as on, \
            _build(product, states, rate_cats, ntips, nsites, tree, False) as off, \
            _build(oracle, states, rate_cats, ntips, nsites, tree) as ref:
        lnl = pc.full_traversal(on)
        st = on.transient_stats()
        if ntips >= 8 and not on.repeat_stats().cherries:
            assert st.skipped > 0, "the resident schedule stored every vector"
        assert st.materialized == 0
        # the likelihood alone touches no vector inside a chain
        assert lnl == pc.full_traversal(off)
        a, b = _everything(on), _everything(off)
        _same(a, b)
        assert on.transient_stats().materialized > 0
        # ... and against the oracle: likelihood, every vector, every scaler count
        lr = pc.full_traversal(ref)
        assert lnl_close(lnl, lr, nsites, states)
        on.tree = tree
        pc.full_traversal(on)
        for op in tree.ops:
            x, y = on.get_clv(op[0]), ref.get_clv(op[0])
            assert site_err(x, y) <= (CLV_SITE_61 if states > 20 else REL_CLV), op
            assert np.array_equal(on.get_scaler(op[1]), ref.get_scaler(op[1]))


@pytest.mark.parametrize("states", [20, 4, 10])
def test_a_vector_that_was_not_stored_is_stored_before_its_inputs_change(product, states):
    """libpll semantics: a vector stays what it was computed as, whatever happens to the P-matrices or tips afterwards"""
    ntips, nsites = 40, 2000
    tree = pc.Tree(ntips, 42, 43)
    with _build(product, states, 4, ntips, nsites, tree, True) as on, _build(product, states, 4, ntips, nsites, tree, False) as off:
        for inst in (on, off):
            pc.full_traversal(inst)
        assert on.transient_stats().skipped > 0
        inner = [op[0] for op in tree.ops]
        # (1) new lengths for every branch: the vectors are the old ones
        for inst in (on, off):
            inst.update_pmatrices(np.arange(tree.nedges), tree.brlens * 1.7)
        for i in inner:
            assert np.array_equal(on.get_clv(i), off.get_clv(i)), i
        assert on.transient_stats().materialized == on.transient_stats().skipped
        # ... and an evaluation right after new lengths gives the old vectors up without recomputing them
        m0 = on.transient_stats().materialized
        for inst in (on, off):
            pc.full_traversal(inst)
            inst.update_pmatrices(np.arange(tree.nedges), tree.brlens * 0.6, one_by_one=True)
            pc.full_traversal(inst)
        assert on.transient_stats().materialized == m0
        sa, sb = tree.scaler_of(tree.root_a), tree.scaler_of(tree.root_b)
        # (the edge likelihood first: it launches the queued matrices of the last update)
        assert on.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix) == off.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix)
        for i in inner:
            assert np.array_equal(on.get_clv(i), off.get_clv(i)), i
        # (2) a tip changes after the next evaluation
        for inst in (on, off):
            pc.full_traversal(inst)
            cmap = pc.state_charmap(states)
            inst.set_tip_states(3, cmap, (pc.random_codes(ntips, nsites, states, 77)[3] + 48).tobytes())
        for i in inner:
            assert np.array_equal(on.get_clv(i), off.get_clv(i)), i
        # (3) a later list overwrites children of vectors that were not stored: evaluation from another root edge
        for k in (tree.nedges // 2, 1):
            for inst in (on, off):
                t0 = pc.Tree(ntips, 42, 43)
                inst.tree = t0
                pc.full_traversal(inst)
                t2 = pc.Tree(ntips, 42, 43)
                t2.set_root_edge(k)
                # the operations of the new orientation that differ from the old one, as a partial traversal would issue them
                old = {(o[0], frozenset((o[2], o[5]))) for o in t0.ops}
                part = [o for o in t2.ops if (o[0], frozenset((o[2], o[5]))) not in old]
                inst.update_partials(part)
            for i in inner:
                assert np.array_equal(on.get_clv(i), off.get_clv(i)), (k, i)


def test_discarded_vectors_cost_nothing(product):
    ntips, nsites = 50, 3000
    tree = pc.Tree(ntips, 42, 43)
    with _build(product, 20, 4, ntips, nsites, tree, True) as on, _build(product, 20, 4, ntips, nsites, tree, False) as off:
        la = lb = None
        for step in range(4):
            for inst in (on, off):
                # a model-optimisation step: new parameters, every P-matrix, the whole tree, the likelihood
                inst.discard_transient()
                rates = product.gamma_cats(0.4 + 0.2 * step, 4)
                inst.set_model(*pc.protein_model(), rates)
            la, lb = pc.full_traversal(on), pc.full_traversal(off)
            assert la == lb
        st = on.transient_stats()
        assert st.materialized == 0 and st.skipped > 0 and st.discarded == 3 * st.skipped // 4
        # the optimiser is done: everything is there for whoever reads next
        on.set_transient(False)
        sa, sb = tree.scaler_of(tree.root_a), tree.scaler_of(tree.root_b)
        for i in [op[0] for op in tree.ops]:
            assert np.array_equal(on.get_clv(i), off.get_clv(i))
        assert on.edge_lnl(tree.root_a, sa, tree.root_b, sb, tree.root_matrix) == lb


@pytest.mark.parametrize("states", [20, 4])
def test_transient_partitions_share_launches(product, states):
    """pllhip_update_partials_batch with the mode on in some of the partitions"""
    ntips = 30
    tree = pc.Tree(ntips, 42, 43)
    sizes = [900, 2100, 1300]
    ops = tree.ops_with_scalers(True)
    with _build(product, states, 4, ntips, sizes[0], tree, True) as a, _build(product, states, 4, ntips, sizes[1], tree, False, seed_shift=1) as b, \
            _build(product, states, 4, ntips, sizes[2], tree, True, seed_shift=2) as c:
        group = [a, b, c]
        ref = [_build(product, states, 4, ntips, n, tree, False, seed_shift=i) for i, n in enumerate(sizes)]
        try:
            for inst in group + ref:
                inst.update_pmatrices(np.arange(tree.nedges), tree.brlens)
            pc.update_partials_batch(product, group, ops)
            for r in ref:
                r.update_partials(ops)
            assert a.transient_stats().skipped > 0 and c.transient_stats().skipped > 0
            assert b.transient_stats().skipped == 0 or common.FORCED_TRANSIENT
            for x, r in zip(group, ref):
                for op in tree.ops:
                    assert np.array_equal(x.get_clv(op[0]), r.get_clv(op[0]))
                    assert np.array_equal(x.get_scaler(op[1]), r.get_scaler(op[1]))
        finally:
            for r in ref:
                r.close()


@pytest.mark.parametrize("states", [20, 4])
@pytest.mark.parametrize("mode", [1, 2])
def test_the_driver_s_full_evaluations(product, oracle, states, mode):
    """pllhip_eval_set_transient: a run of full evaluations under changing parameters (what a model optimiser does),
    then branch-length optimisation, an incremental evaluation from another root and an SPR round on the same
    evaluator -- every number equal to the mode being off, and the likelihoods equal to the oracle's"""
    ntips, nsites = 24, 1500
    tree = pc.Tree(ntips, 42, 43)
    codes = pc.simulated_codes(tree, nsites, states, 45)
    subst, freqs = (pc.protein_model() if states == 20 else (pc.DNA_GTR_RATES, pc.DNA_FREQS))
    out = {}
    for name, lib, m in (("on", product, mode), ("off", product, 0), ("ref", oracle, 0)):
        with pc.Evaluation(lib, tree.newick()) as ev:
            inst = ev.add_partition(0, states, nsites, 4, codes, subst, freqs, 0.7)
            ev.set_transient(m)
            seq = []
            for alpha in (0.7, 0.5, 0.9, 1.3):
                inst.set_model(subst, freqs, lib.gamma_cats(alpha, 4))
                seq.append(ev.loglh())
            if lib is product:
                seq.append(("skipped", inst.transient_stats().skipped > 0))
            seq.append(ev.optimize_branches(iters=2))
            recs = [r for r in ev.records() if r.contents.next]
            ev.L.pllhip_eval_set_root(ev.ev, recs[len(recs) // 2])
            seq.append(ev.loglh(incremental=True))
            seq.append(ev.loglh())
            seq.append(ev.loglh())
            seq.append(ev.spr_round(radius_max=3, ntopol_keep=3)[0])
            seq.append(ev.newick())
            out[name] = seq
    assert out["on"][4] == ("skipped", True) and (out["off"][4] == ("skipped", False) or common.FORCED_TRANSIENT)
    out["on"].pop(4), out["off"].pop(4)
    assert out["on"] == out["off"]
    for a, b in zip(out["on"][:4], out["ref"][:4]):
        assert lnl_close(a, b, nsites, states)
    assert abs(out["on"][-2] - out["ref"][-2]) < 1e-6 * nsites
