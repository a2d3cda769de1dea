"""pllhip_update_partials_batch: one operation list over several partitions (include/pllhip.h).
What it stores must be what per-partition pll_update_partials calls store, bit for bit -- the
reference walks the partitions one after the other (src/tree/treeinfo.c:1020-1056)."""
import numpy as np
import pytest

import common
import pllhip_ctypes as pc

NONE = pc.PLL_SCALE_BUFFER_NONE


def _members(lib, tree, spec, coded=True, attributes=0):
    """spec: [(states, sites, seed_shift), ...] on one tree"""
    out = []
    for states, sites, shift in spec:
        inst = pc.build_instance(lib, states=states, rate_cats=4, ntips=tree.ntips, nsites=sites, coded=coded,
                                 tree=tree, seed_shift=0, attributes=attributes)
        # different data per member: re-seed the tips
        codes = pc.random_codes(tree.ntips, sites, states, 44 + 101 * shift)
        cmap = pc.state_charmap(states)
        for t in range(tree.ntips):
            inst.set_tip_states(t, cmap, (codes[t] + 48).tobytes())
        inst.tree = tree
        out.append(inst)
    return out


def _state(inst):
    t = inst.tree
    clvs = [inst.get_clv(op[0]) for op in t.ops]
    scs = [inst.get_scaler(op[1]) for op in t.ops]
    lnl = inst.edge_lnl(t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
    return clvs, scs, lnl


def _run(lib, members, batched):
    t = members[0].tree
    for m in members:
        m.update_pmatrices(np.arange(t.nedges), t.brlens)
    if batched:
        pc.update_partials_batch(lib, members, t.ops_with_scalers(True))
    else:
        for m in members:
            m.update_partials(t.ops_with_scalers(True))
    return [_state(m) for m in members]


def _assert_same(a, b):
    for (ca, sa, la), (cb, sb, lb) in zip(a, b):
        assert la == lb
        for x, y in zip(ca, cb):
            assert np.array_equal(x, y)
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y)


def test_batch_on_the_oracle_is_the_plain_loop(oracle):
    tree = pc.Tree(9, 42, 43)
    spec = [(4, 300, 0), (20, 120, 1)]
    a = _run(oracle, _members(oracle, tree, spec), True)
    b = _run(oracle, _members(oracle, tree, spec), False)
    _assert_same(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [
    [(20, 1000, 0), (20, 1000, 1), (20, 1000, 2), (20, 1000, 3)],
    [(20, 3000, 0), (20, 37, 1), (20, 1031, 2)],                       # unequal extents, a ragged last block
    [(4, 5000, 0), (4, 1, 1), (4, 777, 2), (4, 64, 3), (4, 65, 4)],
    [(4, 2000, 0), (20, 500, 1), (4, 900, 2), (20, 700, 3), (61, 100, 4)],   # C4's mix + a family without schedules
    [(10, 800, 0), (10, 300, 1), (16, 500, 2), (16, 100, 3)],
])
@pytest.mark.parametrize("ntips", [14, 60])
def test_batch_stores_what_per_partition_calls_store(product, spec, ntips):
    tree = pc.Tree(ntips, 42, 43)
    got = _members(product, tree, spec)
    want = _members(product, tree, spec)
    try:
        a = _run(product, got, True)
        b = _run(product, want, False)
        _assert_same(a, b)
        # fewer launches than one set per partition: members of a family share theirs
        shared = sum(m.counters().partial_launches for m in got)
        single = sum(m.counters().partial_launches for m in want)
        # (partitions that compute per class of sites keep their own launches: their schedules differ with their data)
        assert shared < single or common.FORCED_REPEATS, (shared, single)
        # a second, different list (another root edge): the merged schedule is rebuilt
        t2 = pc.Tree(ntips, 42, 43)
        t2.set_root_edge(tree.nedges // 3)
        for m in got + want:
            m.tree = t2
        _assert_same(_run(product, got, True), _run(product, want, False))
    finally:
        for m in got + want:
            m.close()


@pytest.mark.gpu
def test_batch_skips_remote_partitions_and_takes_single_operations(product):
    tree = pc.Tree(12, 42, 43)
    spec = [(20, 640, 0), (20, 200, 1), (20, 90, 2)]
    got, want = _members(product, tree, spec), _members(product, tree, spec)
    try:
        for m in got + want:
            m.update_pmatrices(np.arange(tree.nedges), tree.brlens)
        ops = tree.ops_with_scalers(True)
        for op in ops:                      # one operation per call, as the branch-length optimiser issues them
            pc.update_partials_batch(product, [got[0], None, got[1], got[2]], [op])
            for m in want:
                m.update_partials([op])
        _assert_same([_state(m) for m in got], [_state(m) for m in want])
    finally:
        for m in got + want:
            m.close()


@pytest.mark.gpu
def test_batch_with_per_rate_scalers_and_deep_trees(product):
    """scaling really happens (260 taxa) and the per-rate counts travel through the shared launches"""
    tree = pc.Tree(260, 42, 43)
    for attributes in (0, pc.PLL_ATTRIB_RATE_SCALERS):
        spec = [(20, 200, 0), (20, 333, 1)]
        got, want = _members(product, tree, spec, attributes=attributes), _members(product, tree, spec, attributes=attributes)
        try:
            a, b = _run(product, got, True), _run(product, want, False)
            _assert_same(a, b)
            assert max(int(s.max()) for s in a[0][1]) >= 1
        finally:
            for m in got + want:
                m.close()


@pytest.mark.gpu
def test_batch_reports_bad_indices(product):
    tree = pc.Tree(8, 42, 43)
    ms = _members(product, tree, [(20, 100, 0), (20, 100, 1)])
    try:
        with pytest.raises(RuntimeError):
            pc.update_partials_batch(product, ms, [(99, NONE, 0, 0, NONE, 1, 1, NONE)])
    finally:
        for m in ms:
            m.close()
