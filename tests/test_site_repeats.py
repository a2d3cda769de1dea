"""(PLLHIP_CLASS_TABLE_PAIRS=64 in the environment sends every class numbering of this file through the hash table
instead of the table of possible pairs; PLLHIP_SITE_REPEATS=2 runs the whole -m gpu suite under the attribute.)

PLL_ATTRIB_SITE_REPEATS, first step (pll-modules_amd/csrc/kernels_repeats.hpp): cherries are kept per class of
sites (pair of tip codes).  Everything a caller can observe must be identical -- bit for bit -- to the attribute
being off: the reference's own tests run every program with and without it and compare the text
(test/src/common.c:31, test/runtest.py:45-51).

Two checks per case: the engine with the attribute against the engine without it (bit for bit), AND the engine with
the attribute against the CPU oracle on the same inputs (likelihoods, every vector, exact scaler counts, per-site
likelihoods, derivatives) -- an error shared by both device paths would pass the first one only."""
import os

import numpy as np
import pytest

import common
import pllhip_ctypes as pc
from test_gpu_parity import lnl_close, site_err, REL_CLV

pytestmark = pytest.mark.gpu
NONE = pc.PLL_SCALE_BUFFER_NONE
# PLLHIP_SITE_REPEATS=2: every partition is treated as if it had the attribute (tests/test_forced_modes.py runs the
# whole suite that way); "the attribute off" then computes per class as well, and asserts about its statistics are void
FORCED = common.FORCED_REPEATS


def _close_to_oracle(got, ref, nsites, states):
    """`got`: _everything() of the engine, `ref`: of the oracle"""
    assert got.keys() == ref.keys()
    for k in got:
        a, b = got[k], ref[k]
        if k.endswith("scaler"):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), k                      # integers: exact
        elif k.endswith("clv"):
            for x, y in zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b]):
                assert site_err(np.asarray(x), np.asarray(y)) <= REL_CLV, k
        elif k == "persite":
            fin = np.isfinite(b)
            assert np.array_equal(fin, np.isfinite(a))
            assert np.all(np.abs(a[fin] - b[fin]) <= 1e-10 * np.abs(b[fin]) + 1e-11), k
        elif k == "deriv":
            assert np.allclose(np.asarray(a), np.asarray(b), rtol=1e-9, atol=1e-9 * nsites), k
        elif np.isfinite(b):
            assert lnl_close(a, b, nsites, states), (k, a, b)
        else:
            assert a == b, k


def _build(product, tree, nsites, repeats, seed=44, gaps=False, states=20, ambiguity=False, coded=True):
    inst = pc.build_instance(product, states=states, rate_cats=4, ntips=tree.ntips, nsites=nsites, coded=coded, tree=tree,
                             attributes=pc.PLL_ATTRIB_SITE_REPEATS if repeats else 0)
    if gaps or ambiguity:
        cmap = pc.state_charmap(states)
        if ambiguity:
            cmap[ord("B")] = (1 << 2) | (1 << 3)
            cmap[ord("Z")] = (1 << 5) | (1 << 6) if states > 6 else (1 << 1) | (1 << 2)
        codes = pc.random_codes(tree.ntips, nsites, states, seed)
        rnd = pc.splitmix64(seed + 5, tree.ntips * nsites).reshape(tree.ntips, nsites)
        for t in range(tree.ntips):
            seq = (codes[t] + 48).astype(np.uint8)
            seq[rnd[t] % np.uint64(17) == 0] = ord("-")
            if ambiguity:
                seq[rnd[t] % np.uint64(23) == 1] = ord("B")
                seq[rnd[t] % np.uint64(29) == 2] = ord("Z")
            inst.set_tip_states(t, cmap, seq.tobytes())
    inst.tree = tree
    return inst


def _everything(inst):
    t = inst.tree
    out = {"lnl": pc.full_traversal(inst)}
    out["clv"] = [inst.get_clv(op[0]) for op in t.ops]
    out["scaler"] = [inst.get_scaler(op[1]) for op in t.ops]
    sa, sb = t.scaler_of(t.root_a), t.scaler_of(t.root_b)
    out["persite"] = inst.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix, persite=True)[1]
    st = inst.alloc_sumtable()
    inst.update_sumtable(t.root_a, t.root_b, sa, sb, st)
    out["deriv"] = [inst.derivatives(sa, sb, x, st) for x in (0.01, 0.3)]
    inst.free_sumtable(st)
    # a second evaluation (the resident schedule is reused), then the same tree from other root edges: cherries
    # become readers' parents, the former root-side vectors become cherries' consumers
    out["lnl2"] = pc.full_traversal(inst)
    for k in (0, t.nedges // 2, 3):
        t2 = pc.Tree(t.ntips, 42, 43, ladder=getattr(t, "is_ladder", False))
        t2.set_root_edge(k)
        inst.tree = t2
        out[f"root{k}"] = pc.full_traversal(inst)
        out[f"root{k}_clv"] = inst.get_clv(t2.ops[-1][0])
    inst.tree = t
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], list):
            for x, y in zip(a[k], b[k]):
                assert np.array_equal(np.asarray(x), np.asarray(y)), k
        else:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


# (5 and 24 states: the 2 .. 32-state family, one and two row tiles -- the reference's own 5-state test runs under the
# attribute, test/runtest.py:45-51)
@pytest.mark.parametrize("states", [20, 4, 5, 24])
@pytest.mark.parametrize("ntips,nsites,gaps,ambiguity", [(14, 1031, False, False), (40, 5000, True, False),
                                                         (9, 257, True, True), (100, 3333, False, False)])
# coded False: the attribute WITHOUT PLL_ATTRIB_PATTERN_TIP, the combination libpll itself allows and the reference's
# harness runs ("sr" without "tv", test/src/common.c:15-31): tips are vectors, and class nodes from the start
@pytest.mark.parametrize("coded", [True, False])
def test_site_repeats_change_nothing_a_caller_can_see(product, oracle, ntips, nsites, gaps, ambiguity, states, coded):
    tree = pc.Tree(ntips, 42, 43)
    with _build(product, tree, nsites, True, gaps=gaps, ambiguity=ambiguity, states=states, coded=coded) as on, \
            _build(product, tree, nsites, False, gaps=gaps, ambiguity=ambiguity, states=states, coded=coded) as off, \
            _build(oracle, tree, nsites, False, gaps=gaps, ambiguity=ambiguity, states=states, coded=coded) as ref:
        a, b = _everything(on), _everything(off)
        _same(a, b)
        _close_to_oracle(a, _everything(ref), nsites, states)
        st = on.repeat_stats()
        # (without pattern tips a cherry is numbered like any other class node and kept only while its classes are at
        # most a quarter -- 4 states: an eighth -- of the sites: not on 1 031 random sites of 20 states)
        if coded or nsites >= 8 * (states + 3) ** 2:
            assert st.cherries > 0 and st.classes < st.sites or nsites < 500
        assert FORCED or off.repeat_stats().cherries == 0


@pytest.mark.parametrize("states,ntips,nsites", [(4, 40, 60_000), (4, 24, 3000), (20, 30, 40_000), (5, 24, 3000),
                                                 (16, 30, 20_000), (3, 30, 40_000)])
@pytest.mark.parametrize("coded", [True, False])
def test_classes_of_whole_subtrees(product, oracle, states, ntips, nsites, coded):
    """second step: nodes above cherries and tips are kept per class too (pairs of the children's classes, numbered
    on the device).  Sequences simulated along the tree (real repeats), a tip whose sequence changes between two
    evaluations (the class maps above it are made again), evaluations from other root edges (class nodes of earlier
    calls under new parents): everything identical to the attribute being off"""
    tree = pc.Tree(ntips, 42, 43, brlen_range=(0.01, 0.12))
    codes = pc.simulated_codes(tree, nsites, states, seed=45)
    out, stats = [], None
    for repeats, lib in ((True, product), (False, product), (False, oracle)):
        inst = pc.build_instance(lib, states=states, rate_cats=4, ntips=ntips, nsites=nsites, coded=coded, tree=tree,
                                 attributes=pc.PLL_ATTRIB_SITE_REPEATS if repeats else 0, codes=codes)
        inst.tree = tree
        with inst:
            res = _everything(inst)
            cmap = pc.state_charmap(states)
            seq = (codes[3] + 48).astype(np.uint8)
            seq[::5] = 48 + (states - 1)
            inst.set_tip_states(3, cmap, seq.tobytes())
            res["after_change"] = pc.full_traversal(inst)
            res["after_change_clv"] = [inst.get_clv(op[0]) for op in tree.ops[::3]]
            res["after_change_scaler"] = [inst.get_scaler(op[1]) for op in tree.ops]
            if repeats:
                stats = inst.repeat_stats()
            out.append(res)
    _same(out[0], out[1])
    _close_to_oracle(out[0], out[2], nsites, states)
    ncherries = sum(1 for op in tree.ops if op[2] < ntips and op[5] < ntips)
    evaluations = 1 + 1 + 3 + 1                              # _everything: two full + three re-rooted, then one more
    assert stats.cherries > ncherries * evaluations          # more class operations than cherries: deeper nodes too
    assert stats.classes * 4 < stats.sites


def test_site_repeats_dna_one_launch_and_rounds(product):
    """4 states at a size where the whole traversal is one launch (and, forced, by rounds): wide tips in both forms"""
    tree = pc.Tree(30, 42, 43)
    with _build(product, tree, 800_000, True, states=4, gaps=True) as on, \
            _build(product, tree, 800_000, False, states=4, gaps=True) as off:
        for rep in range(2):
            assert pc.full_traversal(on) == pc.full_traversal(off)
        for op in tree.ops[::5]:
            assert np.array_equal(on.get_scaler(op[1]), off.get_scaler(op[1]))
            assert np.array_equal(on.get_clv(op[0]), off.get_clv(op[0]))
        st = on.repeat_stats()
        assert st.cherries > 0 and st.classes * 100 < st.sites


@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("states", [20, 4, 7])
def test_site_repeats_with_scaling_cherries(product, oracle, states, coded):
    """a cherry only ever scales when its entries are exact zeros: pendant branches of length 0 (identity
    matrices) make every site with two different states an all-zero, scaled site.  The scaling decision is taken
    per class, the counts travel per site; identical to the attribute being off (lnL is -inf on both sides)."""
    tree = pc.Tree(24, 42, 43)
    cherries = [op for op in tree.ops if op[2] < tree.ntips and op[5] < tree.ntips]
    assert len(cherries) >= 3
    for k, op in enumerate(cherries):
        if k % 3 != 2:
            tree.brlens[op[3]] = 0.0
            tree.brlens[op[6]] = 0.0 if k % 3 == 0 else 0.05
    # (without pattern tips a cherry needs four times as many sites as it has classes to be kept per class)
    nsites = 700 if coded or states < 20 else 2400
    with _build(product, tree, nsites, True, states=states, coded=coded) as on, \
            _build(product, tree, nsites, False, states=states, coded=coded) as off, \
            _build(oracle, tree, nsites, False, states=states, coded=coded) as ref:
        assert pc.full_traversal(ref) == -np.inf
        for rep in range(2):
            la, lb = pc.full_traversal(on), pc.full_traversal(off)
            assert la == lb == -np.inf
            for op in tree.ops:
                assert np.array_equal(on.get_scaler(op[1]), off.get_scaler(op[1]))
                assert np.array_equal(on.get_clv(op[0]), off.get_clv(op[0]))
                assert np.array_equal(on.get_scaler(op[1]), ref.get_scaler(op[1]))       # the oracle's counts, exactly
                assert site_err(on.get_clv(op[0]), ref.get_clv(op[0])) <= REL_CLV
        assert sum(int(off.get_scaler(op[1]).sum()) for op in cherries) > 0
        assert on.repeat_stats().cherries > 0


@pytest.mark.parametrize("states,ntips", [(20, 260), (4, 900), (12, 300)])
def test_site_repeats_on_deep_trees(product, oracle, states, ntips):
    tree = pc.Tree(ntips, 42, 43)
    with _build(product, tree, 300, True, states=states) as on, _build(product, tree, 300, False, states=states) as off, \
            _build(oracle, tree, 300, False, states=states) as ref:
        la, lb = pc.full_traversal(on), pc.full_traversal(off)
        assert la == lb
        assert lnl_close(la, pc.full_traversal(ref), 300, states)
        top = 0
        for op in tree.ops:
            sa, sb = on.get_scaler(op[1]), off.get_scaler(op[1])
            assert np.array_equal(sa, sb)
            assert np.array_equal(sa, ref.get_scaler(op[1]))
            top = max(top, int(sa.max()))
        assert top >= 1
        for op in tree.ops[:40]:
            assert np.array_equal(on.get_clv(op[0]), off.get_clv(op[0]))


def test_site_repeats_through_the_driver(product, oracle):
    """branch-length optimisation and an SPR round (short operation lists, sumtables at cherries): same results,
    and the oracle's likelihoods and moves"""
    out = []
    for attributes, lib in ((pc.PLL_ATTRIB_SITE_REPEATS, product), (0, product), (0, oracle)):
        truth = pc.Tree(12, 7, 8, brlen_range=(0.03, 0.25))
        start = pc.Tree(12, 11, 12, brlen_range=(0.05, 0.15))
        ev = pc.Evaluation(lib, start.newick(), nparts=1)
        r, f = pc.protein_model()
        ev.add_partition(0, 20, 900, 4, pc.simulated_codes(truth, 900, 20), r, f, 0.8, attributes=attributes)
        with ev:
            l0 = ev.loglh()
            l1 = ev.optimize_branches(1e-4, 10.0, 0.01, 2, -1)
            l2, st = ev.spr_round(radius_max=3, ntopol_keep=3)
            out.append((l0, l1, l2, ev.newick(), st.moves_applied))
    assert out[0] == out[1]
    for a, b in zip(out[0][:3], out[2][:3]):
        assert abs(a - b) <= 1e-6 * 900
    assert out[0][4] == out[2][4]


def test_cherry_built_under_an_older_code_table(product, oracle):
    """a tip takes a new ambiguity code after a traversal (the code table grows); a later partial traversal reads
    the cherries built before that: through their expanded vectors, with the same result as without site repeats"""
    tree = pc.Tree(30, 42, 43)
    out = []
    for repeats, lib in ((True, product), (False, product), (False, oracle)):
        with _build(lib, tree, 900, repeats) as a:
            l0 = pc.full_traversal(a)
            cmap = pc.state_charmap(20)
            cmap[ord("B")] = (1 << 2) | (1 << 3)
            cherry_tips = {op[2] for op in tree.ops if op[2] < 30 and op[5] < 30} | {op[5] for op in tree.ops if op[2] < 30 and op[5] < 30}
            other = next(t for t in range(30) if t not in cherry_tips)
            seq = (a.codes[other] + 48).astype(np.uint8)
            seq[::7] = ord("B")
            a.set_tip_states(other, cmap, seq.tobytes())
            ops = [op for op in tree.ops_with_scalers(True) if not (op[2] < 30 and op[5] < 30)]      # everything but the cherries
            a.update_pmatrices(np.arange(tree.nedges), tree.brlens)
            a.update_partials(ops)
            l1 = a.edge_lnl(tree.root_a, tree.scaler_of(tree.root_a), tree.root_b, tree.scaler_of(tree.root_b), tree.root_matrix)
            out.append((l0, l1, [a.get_clv(op[0]) for op in tree.ops]))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][1] != out[0][0]
    assert all(np.array_equal(x, y) for x, y in zip(out[0][2], out[1][2]))
    assert lnl_close(out[0][0], out[2][0], 900, 20) and lnl_close(out[0][1], out[2][1], 900, 20)
    assert all(site_err(x, y) <= REL_CLV for x, y in zip(out[0][2], out[2][2]))


@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("states", [4, 20, 7])
def test_site_repeats_with_ascertainment_bias(product, oracle, states, coded):
    """the ascertainment-bias columns (one per state, behind the alignment, weight 0) are sites like any other for the
    class maps: corrected lnL, per-site values, scaler counts and the corrected derivatives are those of the attribute
    being off, and the oracle's"""
    tree = pc.Tree(30, 42, 43)
    nsites = 6000
    out = []
    for repeats, lib in ((True, product), (False, product), (False, oracle)):
        inst = pc.build_instance(lib, states=states, rate_cats=4, ntips=30, nsites=nsites, coded=coded, tree=tree,
                                 attributes=pc.PLL_ATTRIB_AB_FLAG | pc.PLL_ATTRIB_AB_LEWIS |
                                 (pc.PLL_ATTRIB_SITE_REPEATS if repeats else 0))
        inst.tree = tree
        with inst:
            rng = np.random.RandomState(3)
            inst.set_pattern_weights(rng.randint(1, 5, size=nsites).astype(np.uint32))
            inst.set_asc(pc.PLL_ATTRIB_AB_LEWIS, None)
            res = _everything(inst)
            if repeats:
                st = inst.repeat_stats()
                assert st.cherries > 0
            out.append(res)
    _same(out[0], out[1])
    _close_to_oracle(out[0], out[2], nsites, states)


@pytest.mark.parametrize("coded", [True, False])
def test_site_repeats_in_a_partition_spread_over_engines(product, oracle, coded):
    """engine-internal sharding (pllhip_set_sharding: one pll_partition_t, one engine per contiguous range of sites -- all
    on device 0 here): every shard keeps its own classes (the tips' classes reach the shards through the router:
    upload_tip_classes), the sums are those of the unsharded partition without the attribute"""
    L = product.lib
    tree = pc.Tree(24, 42, 43)
    nsites = 9000
    codes = pc.simulated_codes(tree, nsites, 4, seed=45)
    kw = dict(states=4, rate_cats=4, ntips=24, nsites=nsites, coded=coded, tree=tree, codes=codes)
    plain = pc.build_instance(product, **kw)
    ref = pc.build_instance(oracle, **kw)
    assert L.pllhip_set_sharding(3, None)
    try:
        shard = pc.build_instance(product, attributes=pc.PLL_ATTRIB_SITE_REPEATS, **kw)
    finally:
        assert L.pllhip_set_sharding(0, None)
    with plain, shard, ref:
        assert L.pllhip_shard_count(shard.p) == 3
        for inst in (plain, shard, ref):
            inst.tree = tree
        a, b, c = pc.full_traversal(shard), pc.full_traversal(plain), pc.full_traversal(ref)
        assert lnl_close(a, b, nsites, 4) and lnl_close(a, c, nsites, 4)
        for op in tree.ops[::4]:
            assert np.array_equal(shard.get_clv(op[0]), plain.get_clv(op[0]))
            assert np.array_equal(shard.get_scaler(op[1]), plain.get_scaler(op[1]))
        assert shard.repeat_stats().cherries > 0


@pytest.mark.parametrize("states", [61, 48, 20, 4])
def test_tips_of_partitions_without_pattern_tips(product, oracle, states):
    """Without PLL_ATTRIB_PATTERN_TIP a tip set through pll_set_tip_states is kept per class of sites next to its vector
    (4 / 20 / 2 .. 32 states: wide tips; 33 .. 64 states: byte codes of the engine's own and the family's lookup
    tables), one set through pll_set_tip_clv is a plain vector -- and a tip that changes from one to the other and back
    is read the right way every time: the oracle's lnL, vectors and scaler counts after every change"""
    nsites = 1200 if states <= 20 else 400
    tree = pc.Tree(14, 42, 43)
    insts = [pc.build_instance(lib, states=states, rate_cats=4, ntips=14, nsites=nsites, coded=False, tree=tree)
             for lib in (product, oracle)]
    with insts[0] as a, insts[1] as b:
        def same():
            la, lb = pc.full_traversal(a), pc.full_traversal(b)
            assert lnl_close(la, lb, nsites, states), (la, lb)
            for op in tree.ops[::3]:
                # (33 .. 64 states: the tolerance of the codon P-matrices, tests/test_gpu_parity.py CLV_SITE_61)
                assert site_err(a.get_clv(op[0]), b.get_clv(op[0])) <= (REL_CLV if states <= 32 else 1e-7)
                assert np.array_equal(a.get_scaler(op[1]), b.get_scaler(op[1]))
            return la
        l0 = same()
        # tip 3 as an arbitrary vector (no state set at all: positive numbers), tip 5 as the 0/1 vectors it had
        rng = np.random.RandomState(7)
        vec = rng.uniform(0.1, 1.0, size=(nsites, states))
        codes = a.codes
        onehot = np.zeros((nsites, states))
        onehot[np.arange(nsites), codes[5]] = 1.0
        for inst in (a, b):
            inst.set_tip_clv(3, vec.ravel())
            inst.set_tip_clv(5, onehot.ravel())
        l1 = same()
        assert l1 != l0
        # ... and back through the states
        cmap = pc.state_charmap(states)
        for inst in (a, b):
            inst.set_tip_states(3, cmap, (codes[3] + 48).astype(np.uint8).tobytes())
        l2 = same()
        for inst in (a, b):
            inst.set_tip_states(5, cmap, (codes[5] + 48).astype(np.uint8).tobytes())
        same()
        assert lnl_close(pc.full_traversal(a), l0, nsites, states)
