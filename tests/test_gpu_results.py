"""Deferred scalar results, several trial lengths per sumtable scan, the single-launch
reduction and the device-side reduce (include/pllhip.h) on the GPU: every one of them must
return what the blocking libpll-style calls return, bit for bit."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pllhip_ctypes as pc
from conftest import ROOT
from test_eval_driver import build, check_speculative_newton

pytestmark = pytest.mark.gpu


def _edge(inst):
    t = inst.tree
    return t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix


@pytest.mark.parametrize("states,rate_cats", [(4, 4), (4, 1), (20, 4), (20, 2), (61, 4), (5, 4), (7, 3)])
def test_multi_length_derivatives(product, oracle, states, rate_cats):
    ntips, nsites = (7, 300) if states > 20 else (10, 1031)
    a = pc.build_instance(product, states=states, rate_cats=rate_cats, ntips=ntips, nsites=nsites, pinv=0.1)
    b = pc.build_instance(oracle, states=states, rate_cats=rate_cats, ntips=ntips, nsites=nsites, pinv=0.1,
                          tree=a.tree)
    with a, b:
        for inst in (a, b):
            inst.L.pll_update_invariant_sites(inst.p)
            pc.full_traversal(inst)
        pcl, psc, ccl, csc, _ = _edge(a)
        sa, sb = a.alloc_sumtable(), b.alloc_sumtable()
        a.update_sumtable(pcl, ccl, psc, csc, sa)
        b.update_sumtable(pcl, ccl, psc, csc, sb)
        ts = [0.2, 1e-4, 0.033, 1.7, 9.0, 0.2, 0.5, 0.01]
        single = [a.derivatives(psc, csc, x, sa) for x in ts]
        for count in range(1, 9):
            df, ddf = a.derivatives_multi(psc, csc, ts[:count], sa)
            for k in range(count):
                # a length's result does not depend on what shares the launch
                assert (df[k], ddf[k]) == single[k], (count, k)
        want = [b.derivatives(psc, csc, x, sb) for x in ts]
        # 61 states: each engine runs its own eigen-solver; tests/test_expm_fixtures.py pins both
        # against an independent matrix exponential, here only the kernels' agreement is at stake
        rel = 1e-9 if states <= 20 else 1e-7
        for (g0, g1), (w0, w1) in zip(single, want):
            assert abs(g0 - w0) <= rel * max(1.0, abs(w0)) and abs(g1 - w1) <= rel * max(1.0, abs(w1))
        with pytest.raises(RuntimeError):
            a.derivatives_multi(psc, csc, ts + [0.3], sa)           # at most 8 per call
        c = a.counters()
        assert c.derivative_points > c.derivative_calls
        a.free_sumtable(sa)
        b.free_sumtable(sb)


def test_speculative_newton_on_gpu(product):
    check_speculative_newton(product)


def _results_group(product, comm, insts, ts):
    """lnL of every partition + derivatives at `ts`, through one result group; returns the
    fetched numbers"""
    L = product.lib
    K = len(ts)
    rs = L.pllhip_results_create(comm, len(insts) * 2 * 8)
    assert rs, product.errmsg
    try:
        for k, inst in enumerate(insts):
            pcl, psc, ccl, csc, m = _edge(inst)
            assert L.pllhip_results_edge_loglikelihood(rs, k, inst.p, pcl, psc, ccl, csc, m, inst.params_p)
        lnl = np.zeros(len(insts))
        assert L.pllhip_results_fetch(rs, 0, len(insts), 0, lnl.ctypes.data_as(pc.c_double_p)), product.errmsg
        tt = pc._f64(ts)
        for k, inst in enumerate(insts):
            _, psc, _, csc, _ = _edge(inst)
            assert L.pllhip_results_derivatives(rs, k * 2 * K, inst.p, psc, csc, tt.ctypes.data_as(pc.c_double_p),
                                                K, inst.params_p, inst.sumtable), product.errmsg
        d = np.zeros(len(insts) * 2 * K)
        assert L.pllhip_results_fetch(rs, 0, len(d), 0, d.ctypes.data_as(pc.c_double_p)), product.errmsg
        # a slot nobody deposited to is the identity of the operation
        extra = np.ones(3)
        assert L.pllhip_results_fetch(rs, 5, 3, 0, extra.ctypes.data_as(pc.c_double_p))
        assert list(extra) == [0.0, 0.0, 0.0]
        return lnl, d.reshape(len(insts), K, 2)
    finally:
        L.pllhip_results_destroy(rs)


def _three_partitions(product):
    insts = []
    for states, R, n in ((4, 4, 3000), (20, 4, 777), (61, 2, 130)):
        inst = pc.build_instance(product, states=states, rate_cats=R, ntips=8, nsites=n)
        pc.full_traversal(inst)
        pcl, psc, ccl, csc, _ = _edge(inst)
        inst.sumtable = inst.alloc_sumtable()
        inst.update_sumtable(pcl, ccl, psc, csc, inst.sumtable)
        insts.append(inst)
    return insts


def _blocking(insts, ts):
    lnl = np.array([i.edge_lnl(*_edge(i)) for i in insts])
    d = np.array([[i.derivatives(_edge(i)[1], _edge(i)[3], x, i.sumtable) for x in ts] for i in insts])
    return lnl, d


@pytest.mark.parametrize("ntrial", [1, 3, 4, 6])
def test_deferred_results_equal_blocking_calls(product, ntrial):
    ts = [0.11, 1e-4, 0.9, 0.35, 2.0, 0.07][:ntrial]
    insts = _three_partitions(product)
    try:
        want = _blocking(insts, ts)
        got = _results_group(product, None, insts, ts)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    finally:
        for i in insts:
            i.free_sumtable(i.sumtable)
            i.close()


def test_deferred_results_through_rccl_world_of_one(product):
    """the communicator path (device slots, ncclAllReduce in place, publish kernel, one wait)
    with the one rank this box has: same numbers as the blocking calls"""
    L = product.lib
    idbuf = C.create_string_buffer(128)
    assert L.pllhip_comm_get_unique_id(idbuf), product.errmsg
    comm = L.pllhip_comm_create(idbuf.raw, 0, 1, 0)
    assert comm, product.errmsg
    insts = _three_partitions(product)
    try:
        assert L.pllhip_comm_size(comm) == 1 and L.pllhip_comm_rank(comm) == 0
        ts = [0.11, 1e-4, 0.9]
        want = _blocking(insts, ts)
        for _ in range(3):                       # sequence words, pending lists and slots are reused
            got = _results_group(product, comm, insts, ts)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # the host-payload callback on the same communicator
        buf = np.array([1.5, -2.0, 7.0])
        L.pllhip_reduce_cb(comm, buf.ctypes.data_as(pc.c_double_p), 3, 0)
        assert list(buf) == [1.5, -2.0, 7.0]
    finally:
        for i in insts:
            i.free_sumtable(i.sumtable)
            i.close()
        L.pllhip_comm_destroy(comm)


def test_driver_on_deferred_results(product):
    """the C driver with a result group attached (what bench.py runs): identical likelihoods and
    branch-length optimisation, one wait per evaluation / Newton round instead of one per partition"""
    L = product.lib
    idbuf = C.create_string_buffer(128)
    assert L.pllhip_comm_get_unique_id(idbuf)
    comm = L.pllhip_comm_create(idbuf.raw, 0, 1, 0)
    assert comm, product.errmsg
    try:
        out = []
        for attach in (None, "local", "rccl"):
            with build(product, ntips=12) as ev:
                if attach:
                    ev.attach_comm(comm if attach == "rccl" else None)
                lnl = ev.loglh()
                opt = ev.optimize_branches(1e-4, 10.0, 0.01, 4, -1)
                out.append((lnl, opt, ev.newick(), ev.counters()[2], ev.newton_iterations()))
        assert out[0] == out[1] == out[2]
    finally:
        L.pllhip_comm_destroy(comm)


def test_single_launch_reduction_equals_two_launch_form(product):
    """PLLHIP_FUSED_FINISH=0 keeps the block totals + k_final_sum form of round 1: same bits"""
    vals = []
    for fused in ("1", "0"):
        os.environ["PLLHIP_FUSED_FINISH"] = fused
        try:
            row = []
            for states, n in ((4, 100003), (20, 40001), (61, 3001), (5, 2000)):
                inst = pc.build_instance(product, states=states, rate_cats=4, ntips=7, nsites=n, pinv=0.05)
                with inst:
                    inst.L.pll_update_invariant_sites(inst.p)
                    row.append(pc.full_traversal(inst))
                    pcl, psc, ccl, csc, _ = _edge(inst)
                    st = inst.alloc_sumtable()
                    inst.update_sumtable(pcl, ccl, psc, csc, st)
                    row.append(inst.derivatives(psc, csc, 0.21, st))
                    row.append(tuple(map(tuple, inst.derivatives_multi(psc, csc, [0.3, 0.01, 5.0], st))))
                    for _ in range(20):          # tickets return to zero after every launch
                        assert inst.derivatives(psc, csc, 0.21, st) == row[-2]
                    inst.free_sumtable(st)
            vals.append(row)
        finally:
            del os.environ["PLLHIP_FUSED_FINISH"]
    assert vals[0] == vals[1]


def test_c_driver_across_two_processes_on_one_gpu(tmp_path):
    """two worker processes share GPU 0, each with its own partitions of the HIP engine (sites
    sharded / partitions distributed with NULL slots), synchronised through the reduce callback:
    the product's N > 1 control flow where only one GPU exists"""
    from test_multirank import _run_driver_workers, check_driver_ranks
    for mode in ("sites", "parts"):
        d = tmp_path / mode
        d.mkdir()
        check_driver_ranks(_run_driver_workers("product", mode, d))


def test_a_failing_rank_takes_its_peers_down_with_it_on_the_gpu(tmp_path):
    """the same with both workers on GPU 0 of the HIP engine (reduce callback over gloo)"""
    from test_multirank import run_fault_workers
    run_fault_workers("product", tmp_path)


@pytest.mark.parametrize("mode", ["deposit", "collective", "publish"])
def test_communicator_failure_paths(tmp_path, mode):
    """the library's own RCCL communicator (a world of one: all this box can hold) with injected failures
    (include/pllhip.h, pllhip_results_fetch):
      deposit     a deferred result cannot be enqueued: that evaluation returns NaN with the cause in pll_errmsg,
                  the collective still runs (with NaN), the group is clean afterwards and the next evaluations work
      collective  the all-reduce fails: ncclCommAbort, NaN, and every later evaluation fails at once (aborted)
      publish     the result never arrives (what a lost peer looks like): the bounded wait gives up after
                  PLLHIP_COLLECTIVE_TIMEOUT_S, the communicator is aborted, later evaluations fail at once"""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_worker.py"), "product", mode,
                          str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, out.stderr[-2000:]
    r = json.load(open(tmp_path / "rank0.json"))
    assert r["lnl"][0] is not None
    assert r["lnl"][1] is None and r["errno"][1] != 0
    if mode == "deposit":
        assert "injected deposit" in r["errmsg"][1]
        assert r["lnl"][2] == r["lnl"][0] and r["lnl"][3] == r["lnl"][0]
    else:
        assert "aborted" in r["errmsg"][1]
        if mode == "publish":
            assert r["errno"][1] == 903                                     # PLL_ERROR_HIP_TIMEOUT
        assert r["lnl"][2] is None and r["lnl"][3] is None
        assert r["errno"][2] == 904 and r["errno"][3] == 904                # PLL_ERROR_HIP_COMM_ABORTED


@pytest.mark.parametrize("states", [20, 4, 61])
def test_device_newton_that_cannot_get_its_workgroups(states):
    """The device-resident Newton-Raphson loop sizes its grid for a device the partition has to itself; a workgroup
    that never arrives (tests/_newton_stall_worker.py) ends the launch with PLLHIP_ERROR_NEWTON_STUCK after the bounded
    wait instead of failing the optimisation: the engine's reductions keep working, the driver redoes the branch on
    the host loop and stays there -- same likelihoods, same tree, same iterate count as without the fault."""
    import json
    import subprocess
    import sys
    got = {}
    for stall in (1, 0):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_newton_stall_worker.py"), str(states), str(stall)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        got[stall] = json.loads(out.stdout.strip().splitlines()[-1])
    # PLLHIP_ERROR_NEWTON_STUCK with the fault; without it the loop ends as the data have it (910: the iteration limit)
    assert got[1]["direct_errno"] == 913 and got[0]["direct_errno"] in (0, 910)
    assert got[1]["deriv_unchanged"] and got[0]["deriv_unchanged"]
    for k in ("lnl_after", "lnl0", "lnl1", "newick", "iterations"):
        assert got[1][k] == got[0][k], k


@pytest.mark.parametrize("states,nshards", [(20, 3), (4, 2), (61, 2), (10, 4)])
def test_partition_spread_over_devices(product, states, nshards):
    """engine-internal sharding (include/pllhip.h: pllhip_set_sharding; SURVEY.md 8e topology i): ONE
    pll_partition_t whose sites are split into contiguous ranges, one engine per range -- on a node
    with several GPUs one per device, here all on device 0.  Everything a caller can observe must
    be what the unsharded partition gives (sums differ in their grouping only)."""
    L = product.lib
    nsites = 1000 if states <= 20 else 400
    kw = dict(states=states, rate_cats=4, ntips=9, nsites=nsites, coded=True, pinv=0.1)
    plain = pc.build_instance(product, **kw)
    assert L.pllhip_set_sharding(nshards, None)
    try:
        shard = pc.build_instance(product, **kw, tree=plain.tree)
    finally:
        assert L.pllhip_set_sharding(0, None)
    with plain, shard:
        assert L.pllhip_shard_count(shard.p) == nshards and L.pllhip_shard_count(plain.p) == 1
        w = (pc.splitmix64(5, nsites) % np.uint64(4)).astype(np.uint32) + 1
        for inst in (plain, shard):
            inst.set_pattern_weights(w)
            assert inst.L.pll_update_invariant_sites(inst.p)
        inv_a = np.ctypeslib.as_array(plain.p.contents.invariant, shape=(nsites,))
        inv_b = np.ctypeslib.as_array(shard.p.contents.invariant, shape=(nsites,))
        assert np.array_equal(inv_a, inv_b)
        la, lb = pc.full_traversal(plain), pc.full_traversal(shard)
        assert abs(la - lb) <= 1e-12 * abs(la)
        t = plain.tree
        edge = (t.root_a, t.scaler_of(t.root_a), t.root_b, t.scaler_of(t.root_b), t.root_matrix)
        (_, pa), (_, pb) = plain.edge_lnl(*edge, persite=True), shard.edge_lnl(*edge, persite=True)
        assert np.array_equal(pa, pb)                       # per-site values are computed site by site
        for op in t.ops[-3:]:
            assert np.array_equal(plain.get_clv(op[0]), shard.get_clv(op[0]))
            assert np.array_equal(plain.get_scaler(op[1]), shard.get_scaler(op[1]))
        assert np.array_equal(plain.get_pmatrix(3), shard.get_pmatrix(3))
        sa, sb = plain.alloc_sumtable(), shard.alloc_sumtable()
        args = (t.root_a, t.root_b, t.scaler_of(t.root_a), t.scaler_of(t.root_b))
        plain.update_sumtable(*args, sa)
        shard.update_sumtable(*args, sb)
        assert np.array_equal(plain.get_sumtable(sa), shard.get_sumtable(sb))
        for x in (0.01, 0.4):
            da, db = plain.derivatives(args[2], args[3], x, sa), shard.derivatives(args[2], args[3], x, sb)
            assert np.allclose(da, db, rtol=1e-12)
        ma, mb = plain.derivatives_multi(args[2], args[3], [0.3, 0.02, 1.0], sa), \
            shard.derivatives_multi(args[2], args[3], [0.3, 0.02, 1.0], sb)
        assert np.allclose(ma, mb, rtol=1e-12)
        # the caller pokes the model arrays of the ONE partition it sees
        for inst in (plain, shard):
            r = np.ctypeslib.as_array(inst.p.contents.rates, shape=(4,))
            r[:] = r * 1.3
        la2, lb2 = pc.full_traversal(plain), pc.full_traversal(shard)
        assert abs(la2 - la) > 1.0 and abs(la2 - lb2) <= 1e-12 * abs(la2)
        assert np.allclose(plain.node_ancestral(*edge), shard.node_ancestral(*edge), rtol=1e-12, atol=1e-300)
        plain.free_sumtable(sa)
        shard.free_sumtable(sb)


def test_driver_on_a_partition_spread_over_devices(product):
    """the evaluation driver (branch-length optimisation, deferred results) on sharded partitions:
    same path, same numbers to rounding"""
    L = product.lib
    out = []
    for shards in (0, 3):
        assert L.pllhip_set_sharding(shards, None)
        try:
            with build(product, ntips=12, sizes=(700, 300)) as ev:
                if shards:
                    assert all(L.pllhip_shard_count(i.p) == 3 for i in ev.parts)
                    ev.attach_comm(None)
                lnl = ev.loglh()
                opt = ev.optimize_branches(1e-4, 10.0, 0.01, 4, -1)
                out.append((lnl, opt, ev.newton_iterations()))
        finally:
            assert L.pllhip_set_sharding(0, None)
    assert abs(out[0][0] - out[1][0]) < 1e-11 * abs(out[0][0])
    assert abs(out[0][1] - out[1][1]) < 1e-8 * abs(out[0][1]) and out[0][2] == out[1][2]


def test_bench_two_ranks_sharing_one_gpu():
    """`bench.py --gpus 2` end to end where only one GPU exists: the launcher starts two ranks
    (torch.distributed.run), both use GPU 0 (PLLHIP_ALLOW_DEVICE_WRAP), each owns half of the sites of ONE
    alignment, and the lnL is reduced through the reference's reduce hook over gloo (`--comm torch`; the
    default `--comm rccl` needs one GPU per rank).  Same lnL as the one-rank run of the same configuration."""
    root = ROOT
    args = ["--config", "c2", "--sites", "30000", "--taxa", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, PLLHIP_ALLOW_DEVICE_WRAP="1", PLLHIP_BENCH_DIST_BACKEND="gloo")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args,
                         check=True, capture_output=True, text=True, timeout=600).stdout
    b = json.loads([ln for ln in one.splitlines() if ln.startswith("{")][-1])
    # --comm torch: the rehearsal path; default (--comm rccl): RCCL refuses two ranks on one device, the
    # ranks agree on that and fall back to the reduce hook -- a failing communicator still gives a line
    for extra in (["--comm", "torch"], []):
        two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + extra + args,
                             env=env, check=True, capture_output=True, text=True, timeout=600)
        assert two.stdout.count("\n") == 1                      # ONE line on stdout
        a = json.loads(two.stdout)
        assert a["n_gpus"] == 2 and b["n_gpus"] == 1 and a["scaling"] == "strong"
        assert a["config"]["sites_total"] == b["config"]["sites_total"] == 30000 and a["config"]["sites_per_gpu"] == 15000
        assert abs(a["lnl"] - b["lnl"]) <= 1e-9 * abs(b["lnl"])
        assert "reduce hook" in a["config"]["workload"]
        if not extra:
            assert "falling back" in two.stderr


def test_bench_c4_partitions_balanced_over_two_ranks():
    """`bench.py --config c4 --gpus 2`: the four partitions are shared out by cost -- whole partitions / large
    slices per rank, NULL slots for the rest -- and the summed lnL is that of the one-rank run"""
    args = ["--config", "c4", "--sites", "40000", "--taxa", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, PLLHIP_ALLOW_DEVICE_WRAP="1", PLLHIP_BENCH_DIST_BACKEND="gloo")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args,
                         check=True, capture_output=True, text=True, timeout=600).stdout
    b = json.loads([ln for ln in one.splitlines() if ln.startswith("{")][-1])
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "torch"] + args,
                         env=env, check=True, capture_output=True, text=True, timeout=600)
    a = json.loads(two.stdout)
    assert a["n_gpus"] == 2 and a["config"]["sites_total"] == b["config"]["sites_total"] == 30000
    assert "cost-balanced" in a["config"]["partition_assignment"]
    parts = a["config"]["partitions"]
    assert any(p.get("remote") for p in parts) and sum(p["sites_on_rank0"] for p in parts) < 30000
    assert abs(a["lnl"] - b["lnl"]) <= 1e-9 * abs(b["lnl"])


def test_engine_on_the_runtime_torch_loaded():
    """a rank of `bench.py --gpus N` imports torch first, so the engine and its communicator run on the
    HIP / RCCL libraries of the torch wheel (same SONAMEs as the system's): torch.cuda
    works next to the engine, a world-of-one communicator reduces a driver's lnL on the device, and the
    numbers are those of a process that never saw torch"""
    code = r"""
import sys, ctypes as C
import torch
torch.cuda.set_device(0)
t = torch.ones(8, device="cuda").sum().item()
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pllhip_ctypes as pc
from test_eval_driver import build
lib = pc.PllLib(pc.PRODUCT_LIB)
L = lib.lib
idbuf = C.create_string_buffer(128)
assert L.pllhip_comm_get_unique_id(idbuf), lib.errmsg
comm = L.pllhip_comm_create(idbuf.raw, 0, 1, 0)
assert comm, lib.errmsg
with build(lib, ntips=12) as ev:
    ev.attach_comm(comm)
    res = "%%.17g %%.17g %%g" %% (ev.loglh(), ev.optimize_branches(1e-4, 10.0, 0.01, 2, -1), t)
L.pllhip_comm_destroy(comm)
maps = open("/proc/self/maps").read()
print("RESULT", res, "torch_hip" if any("torch/lib/libamdhip64" in ln for ln in maps.splitlines()) else "system_hip")
""" % (os.path.dirname(pc.__file__), os.path.join(ROOT, "tests"))
    text = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, timeout=600).stdout
    out = [ln for ln in text.splitlines() if ln.startswith("RESULT")][-1].split()[1:]   # (RCCL prints a banner)
    with build(pc.PllLib(pc.PRODUCT_LIB), ntips=12) as ev:
        want = (ev.loglh(), ev.optimize_branches(1e-4, 10.0, 0.01, 2, -1))
    assert (float(out[0]), float(out[1])) == want and float(out[2]) == 8.0
    assert out[3] == "torch_hip"
