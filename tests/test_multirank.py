"""N>1 path on CPU: two gloo ranks, sites sharded, results combined only through
the reference's reduce-callback interface."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("states", [4, 20])
def test_two_ranks_match_single_rank(oracle, tmp_path, states):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_multirank_worker.py"), str(states), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    # every rank holds the same reduced values (all-reduce semantics, in place)
    for k in ("lnl", "df", "ddf", "max", "min", "wsum"):
        assert r0[k] == r1[k], k
    assert abs(r0["lnl"] - r0["full_lnl"]) < 1e-9 * abs(r0["full_lnl"])
    assert abs(r0["df"] - r0["full_df"]) < 1e-9 * max(1.0, abs(r0["full_df"]))
    assert abs(r0["ddf"] - r0["full_ddf"]) < 1e-9 * max(1.0, abs(r0["full_ddf"]))
    assert r0["max"] == [2.0, 0.0] and r0["min"] == [1.0]
    assert r0["wsum"] == r0["full_wsum"]      # pattern_weight_sum is per slice (treeinfo.c:1166)


def _run_driver_workers(lib, mode, tmp_path, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_evaldriver_worker.py"), lib, mode, str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, env=env, timeout=900, capture_output=True)
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(nproc)]


def check_driver_ranks(ranks):
    r0 = ranks[0]
    single = r0["single"]
    for r in ranks[1:]:
        # all-reduce semantics: every worker holds the same numbers and makes the same decisions
        for k in ("lnl", "lnl_opt", "lnl_after", "newick", "scans", "iterations", "reduce_calls", "payloads",
                  "tree_lengths"):
            assert r[k] == r0[k], k
    assert abs(r0["lnl"] - single["lnl"]) < 1e-9 * abs(single["lnl"])
    # the optimiser follows the same path: sums over the workers differ from the single
    # process only in rounding (1e-13 relative), the Newton iterates likewise
    assert abs(r0["lnl_opt"] - single["lnl_opt"]) < 1e-7 * abs(single["lnl_opt"])
    assert abs(r0["lnl_after"] - single["lnl_after"]) < 1e-7 * abs(single["lnl_after"])
    assert r0["iterations"] == single["iterations"]
    assert r0["reduce_calls"] > 0
    for a, b in zip(r0["tree_lengths"], single["tree_lengths"]):
        assert abs(a - b) < 1e-6
    # one message per reduce: P = 3 lnL values, {df, ddf} of every trial length of a scan, or the
    # single MIN that agrees on the number of trial lengths
    # (unlinked: P lengths through the MAX reduce, 2 P derivatives)
    assert all(n in (1, 3) or n % 2 == 0 for n in r0["payloads"])


@pytest.mark.parametrize("mode", ["sites", "parts", "scaled-sites", "unlinked-parts", "unlinked-sites"])
def test_c_driver_across_two_processes(oracle, tmp_path, mode):
    """the C driver itself (pllhip_eval_set_parallel_context; in "parts" mode with NULL
    partition slots) run by two gloo processes reproduces the single-process evaluation and
    branch-length optimisation"""
    check_driver_ranks(_run_driver_workers("oracle", mode, tmp_path))


def run_fault_workers(lib, tmp_path):
    """two gloo ranks, rank 1's second evaluation fails locally: both ranks see NaN in that call (the failing
    rank takes part in the reduction with NaN), nobody hangs, both exit non-zero; the third evaluation works"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_fault_worker.py"), lib, "eval", str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, env=env, timeout=300, capture_output=True, text=True)
    assert out.returncode != 0
    ranks = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r in ranks:
        assert r["lnl"][0] is not None and r["lnl"][1] is None and r["lnl"][2] is not None, r
        assert r["lnl"][0] == r["lnl"][2]
    assert ranks[0]["lnl"][0] == ranks[1]["lnl"][0]
    assert "injected" in ranks[1]["errmsg"][1]            # the cause stays visible on the rank that failed
    return ranks


def test_a_failing_rank_takes_its_peers_down_with_it(oracle, tmp_path):
    run_fault_workers("oracle", tmp_path)


def test_bench_launcher_starts_the_ranks(tmp_path):
    """`python bench.py --gpus N` without a rank environment starts N ranks itself (child
    process, before any GPU call) and forwards their line; PLLHIP_BENCH_LAUNCH_PROBE=1 makes
    the ranks prove their existence over gloo instead of running the GPU workload"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PLLHIP_BENCH_LAUNCH_PROBE"] = "1"
    for n in (2, 3):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == n and line["ranks"] == list(range(n))
        assert line["local_ranks"] == list(range(n)) and line["distinct_processes"] == n


def test_bench_refuses_a_world_size_mismatch():
    """a rank environment that does not match --gpus is an error, never a line with the wrong n_gpus"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PLLHIP_BENCH_LAUNCH_PROBE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
